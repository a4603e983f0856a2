import ctypes as C, sys
sys.path.insert(0, ".")
from llamole_amd import _lib
lib = _lib.load()
CFG = {2: "pipe64x32s4", 12: "sk32w8", 13: "sk16w8", 14: "sk32w4", 15: "sk16w16", 16: "sk64w8"}
shapes = {"qkv": (3072, 1024), "proj": (1024, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096)}
for name, (N, K) in shapes.items():
    row = []
    for cfg, cn in CFG.items():
        ms = C.c_float()
        nw = max(2, int(400e6 // (N * K * 2)))
        rc = lib.ll_gemm_bench(64, N, K, cfg, 1, 0, 4 * nw, nw, C.byref(ms))
        row.append(f"{cn}={ms.value*1e3:.1f}" if rc == 0 else f"{cn}=ERR")
    print(name, " ".join(row))
