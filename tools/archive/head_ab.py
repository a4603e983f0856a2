#!/usr/bin/env python3
"""Interleaved A/B of a few rows16 geometries on the GIN template head (same-box, repeated: single sweeps are +-7 % noisy)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib
lib = _lib.load()
M, N, K = 16, 180576, 2048
ms = C.c_float()
cfgs = [(256, 4, 2), (256, 8, 2), (128, 4, 4), (256, 8, 4), (512, 4, 2)]
acc = {c: [] for c in cfgs}
for rep in range(6):
    for c in cfgs:
        lib.ll_set_rows16_geometry(*c)
        lib.ll_rows16_bench(M, N, K, 0x100, 0, 40, 2, C.byref(ms))
        acc[c].append(ms.value * 1e3)
lib.ll_set_rows16_geometry(0, 0, 0)
for c in cfgs:
    v = sorted(acc[c])
    print(c, "min %.1f median %.1f max %.1f us" % (v[0], v[len(v) // 2], v[-1]))
