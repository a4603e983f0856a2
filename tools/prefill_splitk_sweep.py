"""Prefill-sized Linears (65..128 token rows): ll_linear in one piece against ll_linear_splitk_bf16 (K split into f32 slabs + slab sum), the
weights of consecutive calls distinct (a model's layers) so that the 256 MB cache does not help.  python tools/prefill_splitk_sweep.py [M]"""
import os
import sys

os.environ.setdefault("LLAMOLE_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llamole_amd import _lib  # noqa: E402

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
st = torch.cuda.current_stream().cuda_stream
SHAPES = {"qwen2-7b": dict(qkv=(4608, 3584), o=(3584, 3584), gate_up=(18944, 3584), down=(3584, 18944)),
          "llama-3.1-8b": dict(qkv=(6144, 4096), o=(4096, 4096), gate_up=(14336, 4096), down=(4096, 14336))}


def timed(fn, n=40):
    for _ in range(5):
        fn(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for model, shapes in SHAPES.items():
    for name, (N, K) in shapes.items():
        nw = max(3, int(600e6 // (N * K * 2)) + 1)
        W = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02 for _ in range(nw)]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ws = torch.empty(16 * M * N, device="cuda", dtype=torch.float32)

        def one(i):
            assert lib.ll_linear(_lib.LL_BF16, x.data_ptr(), K, W[i % nw].data_ptr(), K, None, out.data_ptr(), N, M, N, K, 0, 0, st) == 0

        line = f"{model:13s} {name:8s} M={M} N={N} K={K}: one piece {timed(one):6.1f} us"
        ref = out.clone()
        for splits in (2, 4, 8):
            if K % (splits * 64):
                continue

            def sk(i, splits=splits):
                assert lib.ll_linear_splitk_bf16(x.data_ptr(), K, W[i % nw].data_ptr(), K, None, out.data_ptr(), N, M, N, K, 0, splits, ws.data_ptr(), st) == 0

            line += f" | x{splits} {timed(sk):6.1f}"
        print(line, flush=True)
        del W
