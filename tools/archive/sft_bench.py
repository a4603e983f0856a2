#!/usr/bin/env python3
"""SFT step throughput on one MI355X (BASELINE configs[4] shape): HF causal LM (random-init, bf16) with LoRA
adapters, three trainable connectors, frozen HIP GIN encoder + predictor (forward and reverse sweep), AdamW.  The LLM
forward/backward is stock PyTorch-ROCm autograd; the graph side of the loss runs in libllamole_hip.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import e2e, synth  # noqa: E402
from tools.gin_bench import fast_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--llm", default="mistral-7b")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--out-dim", type=int, default=180576)
    a = ap.parse_args()
    from llamole_amd.graph_data import GraphBatch
    from llamole_amd.graph_encoder import GraphCLIP
    from llamole_amd.graph_predictor import GraphPredictor
    from llamole_amd.modeling_llamole import SPECIAL_TOKENS, GraphLLMForCausalMLM
    from llamole_amd.sft import GraphSFTCollator, add_lora, sft_step, to_device
    dev = torch.device("cuda")
    llm = e2e.build_llm(a.llm, dev, torch.bfloat16)
    n_lora = add_lora(llm)
    L, H = 5, 512
    enc = GraphCLIP(L, H, 0.0, {})
    enc.to(dev)
    enc.molecule_encoder.load_state_dict(fast_weights(synth.gin_weight_shapes(L, H, "encoder"), dev, 1))
    enc.molecule_projection.load_state_dict(fast_weights(synth.proj_weight_shapes(H), dev, 2))
    pred = GraphPredictor(L, H, 0.0, a.out_dim, {}, {})
    pred.to(dev)
    pred.predictor.load_state_dict(fast_weights(synth.gin_weight_shapes(L, H, "predictor", a.out_dim), dev, 3))
    for m in (enc, pred):
        for p in m.parameters():
            p.data = p.data.to(torch.bfloat16)
            p.requires_grad = False
    V = llm.config.vocab_size
    tid = {t: V - 19 + i for i, t in enumerate(SPECIAL_TOKENS)}
    model = GraphLLMForCausalMLM(types.SimpleNamespace(), types.SimpleNamespace(loss_weight_lm=1, loss_weight_design=1, loss_weight_retro=1),
                                 types.SimpleNamespace(learned_query_size=8), llm, types.SimpleNamespace(text_input_size=768), pred, enc, tid, None)
    for nm in ("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor"):
        getattr(model, nm).to(device=dev, dtype=torch.bfloat16)
    x, ei, ea, batch = synth.make_mol_graphs(8, 0, min_atoms=32, max_atoms=32)
    graphs = dict(enumerate(GraphBatch(x, ei, ea, batch, [32] * 8).to_data_list()))
    g = torch.Generator().manual_seed(0)
    feats = []
    for i in range(a.batch):
        ids = torch.randint(5, V - 64, (a.seq,), generator=g).tolist()
        ids[7] = tid["<molecule>"]
        for q, start in enumerate((a.seq // 3, 2 * a.seq // 3)):
            ids[start] = tid["<retro_start>"]
            ids[start + 1:start + 9] = [tid["<retro_body>"]] * 8
        feats.append({"input_ids": ids, "labels": [-100] * 16 + ids[16:], "molecule_ids": [i % 8],
                      "retro_product_ids": [(i + 1) % 8, (i + 2) % 8], "retro_labels": [int(torch.randint(0, a.out_dim, (1,), generator=g)) for _ in range(2)]})
    b = to_device(GraphSFTCollator(0, graphs)(feats), dev)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4)
    for _ in range(a.warmup):
        log = sft_step(model, b, opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        log = sft_step(model, b, opt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # graph side alone: encoder forward + predictor forward / cross-entropy / reverse sweep for the same batch
    rp = b["retro_product_graphs"]
    c = torch.randn(rp.num_graphs, 768, device=dev, dtype=torch.bfloat16)
    lab = b["retro_labels"].flatten()
    mg = b["molecule_graphs"]

    def graph_side():
        enc(mg.x, mg.edge_index, mg.edge_attr, mg.batch)
        cg = c.clone().requires_grad_(True)
        torch.nn.functional.cross_entropy(pred(rp.x, rp.edge_index, rp.edge_attr, rp.batch, cg).float(), lab).backward()
    for _ in range(3):
        graph_side()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        graph_side()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 10
    print(json.dumps({"workload": f"SFT step: {a.llm} architecture (random-init bf16) + LoRA r=8 on {n_lora} Linears, batch {a.batch} x {a.seq} tokens, "
                                  f"1 spliced molecule + 2 retro queries per sample, GIN H=512 L=5, {a.out_dim} templates",
                      "samples_per_s": a.batch / dt, "tokens_per_s": a.batch * a.seq / dt, "step_ms": 1e3 * dt,
                      "graph_side_ms": 1e3 * tg, "graph_side_share": tg / dt, "trainable_params": sum(p.numel() for p in params),
                      "loss": log["loss"], "lm_loss": log["lm_loss"], "retro_loss": log["retro_loss"],
                      "max_memory_gb": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
