#!/usr/bin/env python3
"""Geometry sweep of the weight-streaming MFMA Linear on the GIN template head ([16 x 2048] x [180576 x 2048]^T, f32 out):
bytes of a row per block x waves per workgroup x K split (ll_set_rows16_geometry), us per launch over two weight copies."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib
lib = _lib.load()
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 180576, 2048)
ms = C.c_float()
nbytes = N * K * 2 + M * K * 2 + M * N * 4
res = []
for seg, waves, ks in [(0, 0, 0)] + [(s, w, k) for s in (128, 256, 512) for w in (4, 8) for k in (1, 2, 4, 8) if k <= w]:
    lib.ll_set_rows16_geometry(seg, waves, ks)
    rc = lib.ll_rows16_bench(M, N, K, 0x100, 0, 12, 2, C.byref(ms))
    if rc != 0:
        continue
    res.append((ms.value * 1e3, seg, waves, ks))
    print(f"seg {seg:4d} waves {waves} ksplit {ks}: {ms.value * 1e3:8.2f} us  {nbytes / ms.value / 1e6:7.1f} GB/s", flush=True)
lib.ll_set_rows16_geometry(0, 0, 0)
print("best:", min(res))
