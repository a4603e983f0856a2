"""GPU box: <= 32-row GEMMs over the Qwen2-7B decode shapes: current dispatch (cfg -1) vs 32-row LDS-DMA tile configurations."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib

lib = _lib.load()
H, I = 3584, 18944
ms = C.c_float()
for name, N, K in [("qkv", 4608, H), ("o_proj", H, H), ("gate_up", 2 * I, H), ("down", H, I), ("lm_head", 152064, H)]:
    mb = N * K * 2 / 1e6
    nw = max(2, int(600 / mb) + 1)
    for M in (8, 16, 32):
        out = [f"{name:8s} M={M:2d}"]
        for cfg in (-1, 34, 35, 36, 37, 38, 39, 40):
            rc = lib.ll_gemm_bench(M, N, K, cfg, 1, 0, 3 * nw, nw, C.byref(ms))
            out.append(f" c{cfg}: {ms.value*1e3:6.1f}" if rc == 0 else f" c{cfg}:  fail")
        print(" |".join(out), flush=True)
