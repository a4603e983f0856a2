#!/usr/bin/env python3
"""GraphDiT batch-1 panel GEMMs (M = 64) with HBM-cold weights (rotating over > 256 MiB) vs cache-warm weights (one matrix):
how much of a phase is the first-touch latency of its weight slice?  (planning number for an L2 prefetch of the next phase's weights)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib  # noqa: E402

lib = _lib.load()
for name, N, K, splits in (("qkv", 3072, 1024, 1), ("proj", 1024, 1024, 4), ("fc1", 4096, 1024, 1), ("fc2", 1024, 4096, 4)):
    out = []
    for nw in (64, 1):
        ms = C.c_float()
        lib.ll_gemm_bench(64, N, K, -1, splits, 0, 256, nw, C.byref(ms))
        out.append(ms.value * 1e3)
    print(f"M=64 {name} N={N} K={K} splits={splits}: cold {out[0]:.2f} us, warm {out[1]:.2f} us")
