#!/usr/bin/env python3
"""Sweep of the pipelined-GEMM configurations (ll_gemm_bench cfg ids, gemm.hip g_pipe_cfgs) on the GraphDiT block shapes at a
given token count M = 2*B*N: q|k|v, proj (split-K), fc1, fc2 (split-K).  Prints us and TFLOP/s per configuration."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib  # noqa: E402

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if len(sys.argv) > 3:
    lib.ll_set_gemm_krot(int(sys.argv[3]))
cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1, 0, 1, 5, 6, 7, 8, 9, 11, 17, 18, 19, 20, 21, 22]
H = 1024
shapes = [("qkv", 3 * H, H, 1), ("proj", H, H, 1), ("proj", H, H, 2), ("proj", H, H, 4), ("fc1", 4 * H, H, 1), ("fc2", H, 4 * H, 1),
          ("fc2", H, 4 * H, 2), ("fc2", H, 4 * H, 4)]
for name, N, K, splits in shapes:
    row = []
    for cfg in cfgs:
        ms = C.c_float()
        nw = max(2, int(600e6 // (N * K * 2)))
        rc = lib.ll_gemm_bench(M, N, K, cfg, splits, 0, 4 * nw, nw, C.byref(ms))
        row.append((ms.value * 1e3 if rc == 0 else float("nan"), cfg))
    best = min(row)
    print(f"M={M} {name:5s} N={N} K={K} splits={splits}: " + " ".join(f"{c}:{t:.1f}" for t, c in row) +
          f" | best cfg {best[1]} {best[0]:.2f} us = {2.0 * M * N * K / best[0] / 1e6:.0f} TF", flush=True)
