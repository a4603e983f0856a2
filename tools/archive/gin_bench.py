#!/usr/bin/env python3
"""Throughput of the GIN encoder / predictor HIP path at BASELINE.json configs[2] sizes (16 product graphs of 32 heavy
atoms, H_gin=512, 5 layers, 180576 templates, top-50), next to the CPU oracle on the host cores.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import synth  # noqa: E402


from llamole_amd.workloads import device_gin_weights as fast_weights  # noqa: E402,F401  (tools/retro_bench.py, sft_bench.py import it from here)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=16)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--out-dim", type=int, default=180576)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only", default="all", choices=["all", "encoder", "predictor", "train"], help="time one leg only (profiling)")
    a = ap.parse_args()
    from llamole_amd.graph_encoder import GraphCLIP
    from llamole_amd.graph_predictor import GraphPredictor
    dev = torch.device("cuda")
    L, H, G = a.layers, a.hidden, a.graphs
    x, ei, ea, batch = synth.make_mol_graphs(G, 0, min_atoms=32, max_atoms=32)
    enc = GraphCLIP(L, H, 0.0, {})
    enc.to(dev)
    enc.molecule_encoder.load_state_dict(fast_weights(synth.gin_weight_shapes(L, H, "encoder"), dev, 1))
    enc.molecule_projection.load_state_dict(fast_weights(synth.proj_weight_shapes(H), dev, 2))
    pred = GraphPredictor(L, H, 0.0, a.out_dim, {}, {})
    pred.to(dev)
    sdp = fast_weights(synth.gin_weight_shapes(L, H, "predictor", a.out_dim), dev, 3)
    pred.predictor.load_state_dict(sdp)
    for m in (enc, pred):
        for p in m.parameters():
            p.data = p.data.to(torch.bfloat16)
    xs, eis, eas, bs = x.to(dev), ei.to(dev), ea.to(dev), batch.to(dev)
    bs._ll_num_graphs = G      # what GraphBatch.from_data_list does (the production callers): no read-back of batch[-1] per call
    c = torch.randn(G, 768, device=dev)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.iters

    nan = float("nan")
    t_enc = timeit(lambda: enc(xs, eis, eas, bs)) if a.only in ("all", "encoder") else nan
    t_pred = timeit(lambda: pred.topk_templates(xs, eis, eas, bs, c, 50)) if a.only in ("all", "predictor") else nan
    # SFT: retro cross-entropy forward + reverse sweep w.r.t. the condition (weights frozen)
    labels = torch.randint(0, a.out_dim, (G,), device=dev)

    def train_step():
        cg = c.clone().requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(pred(xs, eis, eas, bs, cg).float(), labels)
        loss.backward()
        return cg.grad
    t_train = timeit(train_step) if a.only in ("all", "train") else nan
    dec_bytes = a.out_dim * 4 * H * 2
    out = {"workload": f"GIN encoder + predictor(top-50 of {a.out_dim}) on {G} graphs x 32 atoms, H={H}, L={L}, bf16",
           "encoder_ms": 1e3 * t_enc, "predictor_topk_ms": 1e3 * t_pred,
           "expansions_per_s": G / t_pred, "decoder_weight_bytes": dec_bytes,
           "sft_retro_fwd_bwd_ms": 1e3 * t_train, "sft_retro_graphs_per_s": G / t_train,
           "sft_head_hbm_frac_if_all_time": 2 * dec_bytes / t_train / 8e12,
           "decoder_hbm_frac_if_all_time": dec_bytes / t_pred / 8e12}
    if not a.no_cpu and a.only == "all":
        from bench import usable_cores
        from oracle import gin_oracle as go
        cores = usable_cores()
        torch.set_num_threads(cores)
        sd_cpu = {k: v.float().cpu() for k, v in sdp.items()}
        cc = c.cpu()
        with torch.no_grad():
            go.predictor_forward(sd_cpu, L, x, ei, ea, batch, cc)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 8.0 and n < 20:
                go.template_topk(go.predictor_forward(sd_cpu, L, x, ei, ea, batch, cc), 50)
                n += 1
            t_cpu = (time.perf_counter() - t0) / n
        cg = cc.clone().requires_grad_(True)
        t0 = time.perf_counter()
        n2 = 0
        while time.perf_counter() - t0 < 8.0 and n2 < 10:
            loss = torch.nn.functional.cross_entropy(go.predictor_forward(sd_cpu, L, x, ei, ea, batch, cg), labels.cpu())
            torch.autograd.grad(loss, cg)
            n2 += 1
        t_cpu_train = (time.perf_counter() - t0) / n2
        out["cpu_baseline"] = {"predictor_topk_ms": 1e3 * t_cpu, "expansions_per_s": G / t_cpu, "cores": cores, "kind": "port",
                               "sample": f"{n} forward+top-k calls / {n2} forward+autograd steps of the fp32 oracle",
                               "sft_retro_fwd_bwd_ms": 1e3 * t_cpu_train, "sft_retro_graphs_per_s": G / t_cpu_train}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
