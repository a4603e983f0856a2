import ctypes as C, sys
sys.path.insert(0, ".")
from llamole_amd import _lib
lib = _lib.load()
CFG = {-1: "dispatch", 23: "R4U4", 24: "R2U4", 25: "R8U2", 26: "R4U8", 27: "R2U8", 28: "R1U8", 29: "R8U4", 30: "R1U16"}
shapes = {"q/o": (3584, 3584), "kv": (512, 3584), "gate/up": (18944, 3584), "down": (3584, 18944), "lm_head": (152064, 3584), "gin_head": (180576, 2048)}
for name, (N, K) in shapes.items():
    row = []
    for cfg, cn in CFG.items():
        ms = C.c_float()
        nw = max(2, min(64, int(600e6 // (N * K * 2))))
        rc = lib.ll_gemm_bench(1, N, K, cfg, 1, 0, 4 * nw, nw, C.byref(ms))
        row.append(f"{cn}={ms.value*1e3:.1f}us/{N*K*2/ms.value/1e9:.2f}" if rc == 0 else f"{cn}=ERR")
    print(name, N, K, " ".join(row))
