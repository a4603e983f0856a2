"""Run only the dominant GEMM (block MLP fc1 at the bench's token count) a few times: target for rocprofv3 --pmc.
usage: gemm_one.py M [N K]   (no torch import: keeps the profiled process minimal)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ms = C.c_float()
nw = 48   # 48 x 8 MB of weights: > 256 MiB Infinity Cache, so every launch streams its weights from HBM
_lib.check(lib.ll_gemm_bench(M, N, K, -1, 1, 0, 2 * nw, nw, C.byref(ms)))
print(f"M={M} N={N} K={K} avg {ms.value*1e3:.2f} us")
