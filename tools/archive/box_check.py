"""GPU box: is this one of the MI355X boxes where the LDS-DMA ring GEMM is slow?  fc1 shape at 512 rows on the ring (tuning-table id 64),
the register-staged kernel (id 41: no LDS-DMA), the packed-weight panel kernel (-2), plus a plain copy bandwidth figure."""
import ctypes as C
import time

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llamole_amd import _lib  # noqa: E402

lib = _lib.load()


def t(M, N, K, cfg, sp=1, nw=40):
    ms = C.c_float()
    rc = lib.ll_gemm_bench(M, N, K, cfg, sp, 0, 160, nw, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")


x = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
y = torch.empty_like(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    y.copy_(x)
torch.cuda.synchronize()
bw = 20 * 2 * x.numel() / (time.perf_counter() - t0) / 1e12
print(f"fc1 [512x1024]x[4096x1024]^T: ring16w {t(512, 4096, 1024, 64):.1f} us, ring8w {t(512, 4096, 1024, 17):.1f}, register-staged {t(512, 4096, 1024, 41):.1f}, "
      f"panel {t(512, 4096, 1024, -2):.1f} | M=128 ring {t(128, 4096, 1024, 64):.1f} panel {t(128, 4096, 1024, -2):.1f} | copy {bw:.2f} TB/s")
