import ctypes as C, sys
sys.path.insert(0, ".")
from llamole_amd import _lib
lib = _lib.load()
CFG = {1: "64x64w4", 0: "128x64w4", 17: "128x64w8", 22: "128x64w8s6", 20: "64x64w8", 6: "128x128w4s3", 18: "128x128w8s3", 19: "128x128w8s4", 21: "256x64w8s3"}
shapes = {"qkv": (3072, 1024), "proj": (1024, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096)}
for M in (512, 2048):
    for name, (N, K) in shapes.items():
        row = []
        for cfg, cn in CFG.items():
            for sp in ((1, 2) if N == 1024 else (1,)):
                ms = C.c_float()
                nw = max(2, int(400e6 // (N * K * 2)))
                rc = lib.ll_gemm_bench(M, N, K, cfg, sp, 1 if sp > 1 else 0, 4 * nw, nw, C.byref(ms))
                row.append(f"{cn}{'/k2' if sp > 1 else ''}={ms.value*1e3:.1f}" if rc == 0 else f"{cn}=ERR")
        print(M, name, " ".join(row))
