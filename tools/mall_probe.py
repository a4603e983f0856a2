"""Does a weight matrix read a few microseconds earlier (by a throw-away `touch` kernel, i.e. through the memory-side Infinity Cache) make the
GEMV that streams it faster?  Per iteration the four decode GEMVs of one Qwen2-7B layer on rotating weight sets (8 x 466 MB, so nothing
survives from the previous use); variant `touch`: a touch of this layer's o_proj weights (+ optionally the head of gate|up) right before the
o_proj GEMV -- where the decode attention (6 us, 28 workgroups, HBM idle) sits in the real token.  Run under rocprofv3 --kernel-trace --stats
for the per-kernel durations; prints HIP-event totals per variant.
usage: python tools/mall_probe.py [extra_MB_of_gate_up]"""
import sys

import torch

sys.path.insert(0, ".")
from llamole_amd import _lib                      # noqa: E402
sys.path.insert(0, "tests")
import test_decode_chain_gpu as T                 # noqa: E402

H, nq, I, nd = T.SHAPES["qwen2-7b"]
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib = _lib.load()
sets = [T._inputs(H, nq, I, nd, seed=s) for s in range(8)]
mk = lambda n: torch.zeros(1, n, dtype=torch.bfloat16, device="cuda")      # noqa: E731
outs = (mk(H), mk(I), mk(H), mk(nd))
sink = torch.zeros(4, dtype=torch.int32, device="cuda")


def layer(t, touch, wgs, nt):
    s = torch.cuda.current_stream().cuda_stream
    h1, act, h2, qkv = outs
    if touch:
        _lib.check(lib.ll_weight_touch_probe(t["wo"].data_ptr(), H * nq * 2, wgs, nt, sink.data_ptr(), s), "touch")
        if extra:
            _lib.check(lib.ll_weight_touch_probe(t["wgu"].data_ptr(), extra << 20, wgs, nt, sink.data_ptr(), s), "touch")
    T._four_launches(lib, t, H, nq, I, nd, True, outs)


def graphed(fn):
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


print(f"no touch                 : {graphed(lambda t: layer(t, False, 0, 0)):7.2f} us per layer", flush=True)
for wgs in (256, 512, 1024):
    for nt in (0, 1):
        us = graphed(lambda t: layer(t, True, wgs, nt))
        print(f"touch wgs={wgs:5d} nt={nt} extra={extra:3d} MB: {us:7.2f} us per layer (touch included)", flush=True)
