"""GPU box: time ll_gemv_fused_bf16 on the Qwen2-7B decode shapes (nt on/off, norm prologue, epilogues) against the
unfused gemv_bf16_kernel (ll_gemm_bench cfg -1).  python tools/gemv_fused_sweep.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib

lib = _lib.load()
H, I, V = 3584, 18944, 152064
shapes = [("qkv", 4608, H, 0, 1), ("o_proj", H, H, 1, 0), ("gate_up", I, H, 2, 1), ("down", H, I, 1, 0), ("lm_head", V, H, 0, 0)]
ms = C.c_float()
for M in (1, 2):
    for name, N, K, epi, norm in shapes:
        rows = 2 * N if epi == 2 else N
        mb = rows * K * 2 / 1e6
        nw = max(2, int(600 / mb) + 1)
        iters = max(20, int(2000 / max(mb / 5.0, 1)))
        _lib.check(lib.ll_gemm_bench(M, rows, K, -1, 1, 0, iters, nw, C.byref(ms)), "ll_gemm_bench")
        base = ms.value
        out = [f"M={M} {name:8s} {mb:7.1f} MB  unfused {base*1e3:7.1f} us {mb/base/1e3:5.2f} TB/s |"]
        for nt in (0, 1):
            _lib.check(lib.ll_gemv_fused_bench(M, N, K, epi, norm, nt, iters, nw, C.byref(ms)), "ll_gemv_fused_bench")
            out.append(f" nt={nt} {ms.value*1e3:7.1f} us {mb/ms.value/1e3:5.2f} TB/s |")
        print("".join(out), flush=True)
