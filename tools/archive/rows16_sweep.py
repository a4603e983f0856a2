"""GPU box: ll_linear_rows16_bf16 (5..16-row weight-streaming MFMA Linear) on the decode shapes -- correctness against an f32
matmul of the same bf16 operands and time against the 32-row LDS-DMA ring GEMM (ll_gemm_bench cfg -1).
python tools/rows16_sweep.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib

lib = _lib.load()
dev = torch.device("cuda", 0)
st = lambda: torch.cuda.current_stream().cuda_stream


def check(M, N, K, epi, geom):
    g = torch.Generator(device=dev).manual_seed(M * 7 + N + K + epi)
    rows = 2 * N if epi == 2 else N
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(rows, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(rows, device=dev, generator=g).float() * 0.1
    r = torch.randn(M, N, device=dev, generator=g).bfloat16()
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    lib.ll_set_rows16_geometry(*geom)
    _lib.check(lib.ll_linear_rows16_bf16(x.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), None, 0.0, r.data_ptr(), N, out.data_ptr(), N, M, N, K, epi, st()),
               "ll_linear_rows16_bf16")
    lib.ll_set_rows16_geometry(0, 0, 0)
    y = (x.float() @ w.float().t() + b).bfloat16().float()
    if epi == 1:
        ref = (r.float() + y).bfloat16()
    elif epi == 2:
        ref = (torch.nn.functional.silu(y[:, :N]).bfloat16().float() * y[:, N:]).bfloat16()
    else:
        ref = y.bfloat16()
    err = (out.float() - ref.float()).abs().max().item()
    scale = ref.float().abs().max().item()
    return err, scale


bad = 0
for M in (1, 5, 8, 13, 16):
    for N, K in ((4608, 3584), (3584, 3584), (3584, 18944), (1000, 512), (24, 96)):
        for epi in (0, 1, 2):
            for ks in ((256, 4, 1), (512, 4, 2), (128, 16, 4), (256, 8, 8), (128, 8, 2), (0, 0, 0)):
                err, scale = check(M, N, K, epi, ks)
                flag = "" if err <= 0.02 * max(scale, 1.0) else "  <-- MISMATCH"
                bad += bool(flag)
                if flag or (M == 8 and ks == (0, 0, 0)):
                    print(f"M={M} N={N} K={K} epi={epi} geometry={ks}: max err {err:.4g} (scale {scale:.3g}){flag}", flush=True)
print("correctness:", "FAILED" if bad else "ok")

H, I, V = 3584, 18944, 152064
GEOMS = [(0, 0, 0), (256, 4, 4), (128, 8, 8), (256, 8, 8), (256, 8, 4), (512, 8, 8), (512, 4, 4), (256, 4, 1), (256, 4, 2), (128, 8, 2)]
ms = C.c_float()
for name, N, K, epi, norm in [("qkv", 4608, H, 0, 1), ("o_proj", H, H, 1, 0), ("gate_up", I, H, 2, 1), ("down", H, I, 1, 0), ("lm_head", V, H, 0, 0),
                              ("l3_gate_up", 14336, 4096, 2, 1), ("l3_down", 4096, 14336, 1, 0)]:
    rows = 2 * N if epi == 2 else N
    mb = rows * K * 2 / 1e6
    nw = max(2, int(600 / mb) + 1)
    iters = max(20, int(2000 / max(mb / 5.0, 1)))
    for M in (8, 16):
        _lib.check(lib.ll_gemm_bench(M, rows, K, -1, 1, 0, iters, nw, C.byref(ms)), "ll_gemm_bench")
        out = [f"{name:10s} M={M:2d} {mb:7.1f} MB | ring GEMM {ms.value*1e3:6.1f} us {mb/ms.value/1e3:5.2f} TB/s |"]
        for ks in GEOMS:
            if ks[0] == 512 and epi == 2 and ks[1] > 4:
                continue          # 48 rows x 528 B x 8 waves > 160 KB of LDS
            lib.ll_set_rows16_geometry(*ks)
            rc = lib.ll_rows16_bench(M, N, K, epi, norm, iters, nw, C.byref(ms))
            out.append(f" {ks[0]}/{ks[1]}x{ks[2]}: {ms.value*1e3:5.1f} |" if rc == 0 else f" {ks[0]}/{ks[1]}x{ks[2]}:  n/a |")
        lib.ll_set_rows16_geometry(0, 0, 0)
        print("".join(out), flush=True)
