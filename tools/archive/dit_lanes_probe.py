"""GPU box: B molecules as ONE batch of B vs `lanes` engine replicas running sub-batches concurrently on their own HIP streams
(each replica replays its own hipGraph).  python tools/dit_lanes_probe.py [B]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from llamole_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("B", type=int, nargs="?", default=8)
a = ap.parse_args()
args = argparse.Namespace(hidden=1024, depth=28, heads=16, T=50, guide=2.0, nodes=32, dtype="bf16")
dev = torch.device("cuda", 0)
B, N = a.B, args.nodes
props, text, n_nodes = synth.make_dit_inputs(B, seed=0, max_node=N, n_nodes_fixed=N)
models = []


def model(i):
    while len(models) <= i:
        models.append(bench.build_model(args, dev)[0])
    return models[i]


def run(lanes, reps=3):
    sub = B // lanes
    best = 1e9
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hs = []
        for l in range(lanes):
            sl = slice(l * sub, (l + 1) * sub)
            hs.append(model(l).generate_graphs_async(props[sl], text[sl], -200.0, n_nodes=n_nodes[sl], seed=7 + l))
        outs = [h.result() for h in hs]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r:
            best = min(best, dt)
    return best, [h.run_ms for h in hs]


for lanes in [1, 2, 4, 8]:
    if B % lanes or lanes > B:
        continue
    dt, ms = run(lanes)
    print(f"B={B} lanes={lanes} sub-batch={B // lanes}: {dt * 1e3:7.1f} ms total = {B / dt:6.1f} molecules/s, "
          f"{args.T * B / lanes / dt:7.0f} batch-steps/s per lane, lane trajectory ms {[round(m, 1) for m in ms]}", flush=True)
