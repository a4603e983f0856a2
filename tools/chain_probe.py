"""Decode chain (ll_decode_chain_bf16) against the four ll_gemv_fused_bf16 launches it replaces, back to back on rotating layers' weights
(HBM-resident stream: 8 weight sets > L2 + MALL), HIP events over `iters` layers; every hand-over tuning of ll_set_chain_tuning.
usage: python tools/chain_probe.py [qwen2-7b|llama-3.1-8b]"""
import sys

import torch

sys.path.insert(0, ".")
from llamole_amd import _lib                      # noqa: E402
sys.path.insert(0, "tests")
import test_decode_chain_gpu as T                 # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "qwen2-7b"
H, nq, I, nd = T.SHAPES[name]
lib = _lib.load()
sets = [T._inputs(H, nq, I, nd, seed=s) for s in range(8)]
ctr = torch.zeros(8192, dtype=torch.int32, device="cuda")
mk = lambda n: torch.zeros(1, n, dtype=torch.bfloat16, device="cuda")      # noqa: E731
outs = (mk(H), mk(I), mk(H), mk(nd))
bytes_layer = 2 * (H * nq + 2 * I * H + H * I + nd * H)


def timed(fn, iters=120):
    for i in range(16):
        fn(sets[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(sets[i % 8])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def graphed(fn, iters=400):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(8):
            fn(sets[i % 8])
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(24):
                fn(sets[i % 8])
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (8 * 24) * 1e3


four = lambda t: T._four_launches(lib, t, H, nq, I, nd, True, outs)      # noqa: E731
chain = lambda t: T._chain(lib, t, H, nq, I, nd, True, ctr, outs)      # noqa: E731
us = timed(four)
print(f"{name}: {bytes_layer / 1e6:.1f} MB of weights per layer")
print(f"four launches      : {us:7.1f} us eager  {graphed(four):7.1f} us graphed   ({bytes_layer / us / 1e6:.2f} TB/s eager)")
for sleep, mode, what in ((1, 0, "default"), (8, 0, "sleep 8"), (32, 0, "sleep 32"), (1, 1, "fan-in"), (8, 1, "fan-in sleep 8"), (32, 1, "fan-in sleep 32"),
                          (1, 2, "NO WAIT"), (1, 6, "NO WAIT NO SIGNAL"), (1, 4, "NO SIGNAL (waits time out)")[:8]):
    lib.ll_set_chain_tuning(sleep, mode)
    ctr.zero_()
    us = timed(chain)
    print(f'  [{what}: eager {us:.1f} us err={T._error(lib, ctr)}]', flush=True)
    ug = graphed(chain)
    print(f"chain {what:22s}: {us:7.1f} us eager  {ug:7.1f} us graphed   ({bytes_layer / ug / 1e6:.2f} TB/s graphed)  err={T._error(lib, ctr)}")
lib.ll_set_chain_tuning(1, 0)
