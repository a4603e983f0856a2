import ctypes as C, sys
sys.path.insert(0, ".")
from llamole_amd import _lib
lib = _lib.load()
CFG = {-1: "dispatch", 17: "128x64w8", 20: "64x64w8", 1: "64x64w4", 2: "64x32w4", 3: "64x32w4s8", 4: "64x64w4s8"}
for (M, N, K) in [(16, 180576, 2048), (1, 180576, 2048)]:
    row = []
    for cfg, cn in CFG.items():
        ms = C.c_float()
        rc = lib.ll_gemm_bench(M, N, K, cfg, 1, 1, 6, 2, C.byref(ms))
        row.append(f"{cn}={ms.value*1e3:.0f}us({N*K*2/ms.value/1e9:.2f}TB/s)" if rc == 0 else f"{cn}=ERR")
    print(M, N, K, " ".join(row))
