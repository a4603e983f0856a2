#!/usr/bin/env python3
"""Sweep pipelined-GEMM tile/stage/split configurations on the sampler's shapes (run on the GPU box)."""
import ctypes as C
import sys
sys.path.insert(0, ".")
from llamole_amd import _lib

lib = _lib.load()
CFG = ["128x64s4", "64x64s4", "64x32s4", "64x32s8", "64x64s8", "128x64s6", "128x128s3", "128x128s4", "64x128s4",
       "256x64s3", "64x64s2", "64x128s6", "sk32w8", "sk16w8", "sk32w4", "sk16w16", "sk64w8"]
shapes = {"qkv": (3072, 1024), "proj": (1024, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096)}
for M in [int(a) for a in sys.argv[1:]] or [64, 512]:
    for name, (N, K) in shapes.items():
        res = []
        for cfg in range(-1, len(CFG)):
            for sp in (1, 2, 4, 8):
                if K % (64 * sp) or (K // sp) < 128:
                    continue
                if cfg >= 12 and sp > 1:
                    continue
                ms = C.c_float()
                nw = max(2, int(400e6 // (N * K * 2)))
                rc = lib.ll_gemm_bench(M, N, K, cfg, sp, 1 if sp > 1 else 0, 4 * nw, nw, C.byref(ms))
                if rc != 0:
                    continue
                res.append((ms.value * 1e3, "disp" if cfg < 0 else CFG[cfg], sp))
        res.sort()
        gb = N * K * 2 / 1e9
        fl = 2.0 * M * N * K / 1e12
        best = ", ".join(f"{n}/k{sp}={us:.1f}us" for us, n, sp in res[:6])
        disp = [r for r in res if r[1] == "disp" and r[2] == 1][0][0]
        print(f"M={M} {name} N={N} K={K}: dispatch(k1)={disp:.1f}us | best: {best} | best {gb/res[0][0]*1e6/1e3:.2f} TB/s {fl/res[0][0]*1e6:.0f} TF")
