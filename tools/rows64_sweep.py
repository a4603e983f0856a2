"""GPU box: ll_linear_rows64_bf16 (17..64-row weight-streaming MFMA Linear) on the decode shapes of Llama-3.1-8B / Qwen2-7B -- time per
workgroup geometry against the LDS-DMA ring GEMM (ll_gemm_bench cfg -1) and rows16 at 16 rows.  python tools/rows64_sweep.py [M ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LLAMOLE_TUNING", "1")      # micro-benchmark entry points: the LL_TUNING=1 build
from llamole_amd import _lib

lib = _lib.load()
Ms = [int(a) for a in sys.argv[1:]] or [32, 64]
KSG = [0, 1, 2, 4, 8]
ms = C.c_float()
SHAPES = [("l3_qkv", 6144, 4096, 0, 0), ("l3_o_proj", 4096, 4096, 1, 1), ("l3_gate_up", 14336, 4096, 2, 0), ("l3_down", 4096, 14336, 1, 1),
          ("l3_lm_head", 128256, 4096, 0, 0),
          ("q2_qkv", 4608, 3584, 0, 0), ("q2_o_proj", 3584, 3584, 1, 1), ("q2_gate_up", 18944, 3584, 2, 0), ("q2_down", 3584, 18944, 1, 1)]
for name, N, K, epi, norm in SHAPES:
    rows = 2 * N if epi == 2 else N
    mb = rows * K * 2 / 1e6
    nw = max(2, int(600 / mb) + 1)
    iters = max(20, int(2000 / max(mb / 5.0, 1)))
    for M in Ms:
        _lib.check(lib.ll_gemm_bench(M, rows, K, -1, 1, 0, iters, nw, C.byref(ms)), "ll_gemm_bench")
        out = [f"{name:11s} M={M:2d} {mb:7.1f} MB | ring {ms.value*1e3:6.1f} us {mb/ms.value/1e3:5.2f} TB/s |"]
        best = None
        for ks in KSG:
            if epi == 2 and ks > 1:
                continue
            lib.ll_set_rows64_ksplit(ks)
            rc = lib.ll_rows64_bench(M, N, K, epi, norm, iters, nw, C.byref(ms))
            out.append(f" ksg {ks}: {ms.value*1e3:5.1f} |" if rc == 0 else f" ksg {ks}:  n/a |")
            if rc == 0 and (best is None or ms.value < best[0]):
                best = (ms.value, ks)
        lib.ll_set_rows64_ksplit(0)
        if best:
            out.append(f" best ksg {best[1]} {mb/best[0]/1e3:5.2f} TB/s" + (" (+ output RMSNorm)" if norm else ""))
        print("".join(out), flush=True)
