import ctypes as C, sys, time
sys.path.insert(0, ".")
from llamole_amd import _lib
lib = _lib.load()
def bench(M, N, K, cfg, sp=1, iters=None):
    ms = C.c_float()
    nw = max(2, int(400e6 // (N * K * 2)))
    rc = lib.ll_gemm_bench(M, N, K, cfg, sp, 1 if sp > 1 else 0, iters or 4 * nw, nw, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else -1
print("cold  fc1 M=64 pipe64x32:", bench(64, 4096, 1024, 2), " sk16w8:", bench(64, 4096, 1024, 13))
t = time.time()
while time.time() - t < 3.0:
    big = bench(4096, 4096, 4096, 0, iters=40)
print("big GEMM 4096^3 us:", big, "TF:", 2 * 4096**3 / big / 1e6)
print("warm  fc1 M=64 pipe64x32:", bench(64, 4096, 1024, 2), " sk16w8:", bench(64, 4096, 1024, 13))
print("warm  fc1 M=64 pipe64x32 long run (40k iters):", bench(64, 4096, 1024, 2, iters=40000))
print("warm  fc1 M=512 64x64:", bench(512, 4096, 1024, 1), " 128x64:", bench(512, 4096, 1024, 0))
