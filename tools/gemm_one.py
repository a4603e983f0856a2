"""Run only one dominant kernel a few times: target for rocprofv3 --pmc (no torch import: keeps the profiled process minimal).
usage: gemm_one.py M [N K]                     ll_linear dispatch (GraphDiT fc1 GEMM / unfused LLM GEMV)
       gemm_one.py fused M N K EPI NORM        ll_gemv_fused_bf16 (EPI 0 plain | 1 residual | 2 silu_mul over 2N rows)
       gemm_one.py rows16 M N K EPI NORM       ll_linear_rows16_bf16 (same arguments, M <= 16)
       gemm_one.py rows64 M N K EPI NORM       ll_linear_rows64_bf16 on packed weights (M <= 64; NORM & 1 output pre-norm, & 2 input row scale)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LLAMOLE_TUNING", "1")      # micro-benchmark entry points: the LL_TUNING=1 build
from llamole_amd import _lib
lib = _lib.load()
ms = C.c_float()
if len(sys.argv) > 1 and sys.argv[1] == "fused":
    M, N, K, epi, norm = (int(a) for a in sys.argv[2:7])
    rows = 2 * N if epi == 2 else N
    nw = max(2, int(600e6 // (rows * K * 2)))   # > 256 MiB Infinity Cache: every launch streams its weights from HBM
    _lib.check(lib.ll_gemv_fused_bench(M, N, K, epi, norm, 1, 8 * nw, nw, C.byref(ms)))
    print(f"fused M={M} N={N} K={K} epi={epi} norm={norm} avg {ms.value*1e3:.2f} us")
elif len(sys.argv) > 1 and sys.argv[1] == "rows64":
    M, N, K, epi, norm = (int(a) for a in sys.argv[2:7])
    rows = 2 * N if epi == 2 else N
    nw = max(2, int(600e6 // (rows * K * 2)))
    _lib.check(lib.ll_rows64_bench(M, N, K, epi, norm, 8 * nw, nw, C.byref(ms)))
    print(f"rows64 M={M} N={N} K={K} epi={epi} norm={norm} avg {ms.value*1e3:.2f} us")
elif len(sys.argv) > 1 and sys.argv[1] == "rows16":
    M, N, K, epi, norm = (int(a) for a in sys.argv[2:7])
    rows = 2 * N if epi == 2 else N
    nw = max(2, int(600e6 // (rows * K * 2)))
    _lib.check(lib.ll_rows16_bench(M, N, K, epi, norm, 8 * nw, nw, C.byref(ms)))
    print(f"rows16 M={M} N={N} K={K} epi={epi} norm={norm} avg {ms.value*1e3:.2f} us")
else:
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    nw = 48   # 48 x 8 MB of weights: > 256 MiB Infinity Cache, so every launch streams its weights from HBM
    _lib.check(lib.ll_gemm_bench(M, N, K, -1, 1, 0, 2 * nw, nw, C.byref(ms)))
    print(f"M={M} N={N} K={K} avg {ms.value*1e3:.2f} us")
