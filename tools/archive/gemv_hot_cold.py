"""GPU box: ll_gemv_fused_bf16 on one weight matrix re-read every launch (Infinity-Cache / TLB hot) vs cycling through
> 600 MB of distinct matrices (HBM cold): how much of a decode GEMV's time is first-byte latency rather than streaming."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib

lib = _lib.load()
H, I = 3584, 18944
ms = C.c_float()
for name, N, K, epi, norm in [("qkv", 4608, H, 0, 1), ("o_proj", H, H, 1, 0), ("gate_up", I, H, 2, 1), ("down", H, I, 1, 0)]:
    rows = 2 * N if epi == 2 else N
    mb = rows * K * 2 / 1e6
    out = [f"{name:8s} {mb:7.1f} MB"]
    for nw in (1, 2, 4, max(2, int(600 / mb) + 1)):
        _lib.check(lib.ll_gemv_fused_bench(1, N, K, epi, norm, 1, 400, nw, C.byref(ms)), "bench")
        out.append(f" nw={nw:3d}: {ms.value*1e3:6.1f} us")
    print(" |".join(out), flush=True)
