#!/usr/bin/env python3
"""Retrosynthesis expansions on one MI355X at BASELINE configs[2] shape (Qwen2-7B architecture, GIN predictor over 180 576
templates, 16 target molecules), sequential searches (the reference's loop, modeling_llamole.py:1173-1190) vs lock-step
searches with batched expansions and value estimates.  Chemistry is synthetic -- rdkit / rdchiral are not
in this image: products map to seeded 32-atom graphs and templates to scripted reactant strings -- so the numbers are the
GPU side of an expansion (GIN encode, LLM decode, query forward, predictor + top-k, value forward).  One JSON line."""
import argparse
import json
import os
import sys
import time
import types
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import e2e, synth  # noqa: E402
from tools.gin_bench import fast_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--llm", default="qwen2-7b")
    ap.add_argument("--targets", type=int, default=16)
    ap.add_argument("--iterations", type=int, default=2)
    ap.add_argument("--new-tokens", type=int, default=64, help="analysis tokens per expansion (the reference allows 512)")
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--out-dim", type=int, default=180576)
    a = ap.parse_args()
    from llamole_amd.graph_data import GraphBatch
    from llamole_amd.graph_encoder import GraphCLIP
    from llamole_amd.graph_predictor import GraphPredictor
    dev = torch.device("cuda")
    llm = e2e.build_llm(a.llm, dev, torch.bfloat16)
    L, H = 5, 512
    enc = GraphCLIP(L, H, 0.0, {})
    enc.to(dev)
    enc.molecule_encoder.load_state_dict(fast_weights(synth.gin_weight_shapes(L, H, "encoder"), dev, 1))
    enc.molecule_projection.load_state_dict(fast_weights(synth.proj_weight_shapes(H), dev, 2))
    pred = GraphPredictor(L, H, 0.0, a.out_dim, {}, {i: f"T{i}" for i in range(a.out_dim)})
    pred.to(dev)
    pred.predictor.load_state_dict(fast_weights(synth.gin_weight_shapes(L, H, "predictor", a.out_dim), dev, 3))
    for m in (enc, pred):
        for p in m.parameters():
            p.data = p.data.to(torch.bfloat16)
    # scripted chemistry: every template yields one two-reactant outcome derived from (template, product)
    pred.template_runner = lambda t, s: [f"M{zlib.crc32((t + s).encode()) % 997}.M{zlib.crc32((s + t).encode()) % 997}"]
    orch, tok = e2e.build_orchestrator(llm, types.SimpleNamespace(text_input_size=768, check_valid=lambda s: True), dev)
    orch.graph_predictor, orch.graph_encoder = pred, enc
    orch.graph_to_lm_connector = torch.nn.Sequential(torch.nn.Linear(H, llm.config.hidden_size), torch.nn.SiLU()).to(dev, torch.bfloat16)
    x, ei, ea, batch = synth.make_mol_graphs(64, 0, min_atoms=32, max_atoms=32)
    pool = GraphBatch(x, ei, ea, batch, [32] * 64).to_data_list()
    orch.smiles_to_graph = lambda s: type(pool[0])(*(t.clone() for t in (lambda g: (g.x, g.edge_index, g.edge_attr))(pool[zlib.crc32(s.encode()) % 64])))
    info = orch.enable_mi355x_decode()
    kw = dict(expansion_topk=a.topk, iterations=a.iterations, starting_mols={"<none>"}, max_planning_time=1e9, rollback=False,
              design_text="Design", do_sample=True, temperature=0.6, top_p=0.9, max_new_tokens=a.new_tokens,
              eos_token_id=[], pad_token_id=tok.pad_token_id)
    orch.retro_max_new_tokens = a.new_tokens
    targets = [f"TARGET{i}" for i in range(a.targets)]

    def run_sequential():
        return [orch.retrosynthesize(None, t, **kw) for t in targets]

    def run_lock_step():
        return orch.retrosynthesize_many([None] * len(targets), targets, **kw)
    res = {}
    for name, fn in (("sequential", run_sequential), ("lock_step", run_lock_step)):
        torch.manual_seed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        res[name] = time.perf_counter() - t0
        assert len(out) == a.targets
    # the reference's own structure -- one search at a time AND one LLM forward per new tree node -- on a 2-target sample
    orch.batch_values = False
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in targets[:2]:
        orch.retrosynthesize(None, t, **kw)
    torch.cuda.synchronize()
    per_node = (time.perf_counter() - t0) / (2 * a.iterations)
    orch.batch_values = True
    n_exp = a.targets * a.iterations
    print(json.dumps({"workload": f"{a.targets} retrosynthesis searches x {a.iterations} expansions, {a.llm} architecture, analysis <= "
                                  f"{a.new_tokens} tokens per expansion, top-{a.topk} of {a.out_dim} templates, synthetic chemistry",
                      "sequential_s": res["sequential"], "lock_step_s": res["lock_step"],
                      "expansions_per_s_sequential": n_exp / res["sequential"], "expansions_per_s_lock_step": n_exp / res["lock_step"],
                      "speedup": res["sequential"] / res["lock_step"],
                      "expansions_per_s_per_node_values": 1.0 / per_node, "per_node_values_sample": "2 targets",
                      "speedup_vs_per_node_values": per_node * n_exp / res["lock_step"], "llm_acceleration": info}))


if __name__ == "__main__":
    main()
