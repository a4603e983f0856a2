"""BASELINE configs[2] on the GPU box: 16 retrosynthesis searches advanced in lock step, depth <= 5, every round's
expansions as ONE batched LLM decode + GIN encoder forward + full-size predictor forward (H=512, L=5, 180 576 templates) + top-50.
The LLM is the tiny Qwen2 architecture (the decode kernels are covered at full size elsewhere); chemistry is scripted (no rdkit /
rdchiral in the images): a template applied to a product yields two reactants, purchasable with a depth-dependent probability."""
import types
import zlib

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_lock_step_retrosynthesis_batch16_depth5():
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from llamole_amd.workloads import device_gin_weights as fast_weights
    from llamole_amd import e2e, synth
    from llamole_amd.graph_data import GraphBatch
    from llamole_amd.graph_encoder import GraphCLIP
    from llamole_amd.graph_predictor import GraphPredictor
    from llamole_amd.llm_accel import restore_elementwise
    dev = torch.device("cuda")
    llm = e2e.build_llm("tiny", dev, torch.bfloat16)
    L, H, D = 5, 512, 180576
    enc = GraphCLIP(L, H, 0.0, {})
    enc.to(dev)
    enc.molecule_encoder.load_state_dict(fast_weights(synth.gin_weight_shapes(L, H, "encoder"), dev, 1))
    enc.molecule_projection.load_state_dict(fast_weights(synth.proj_weight_shapes(H), dev, 2))
    pred = GraphPredictor(L, H, 0.0, D, {}, {i: f"T{i}" for i in range(D)})
    pred.to(dev)
    pred.predictor.load_state_dict(fast_weights(synth.gin_weight_shapes(L, H, "predictor", D), dev, 3))
    for m in (enc, pred):
        for p in m.parameters():
            p.data = p.data.to(torch.bfloat16)
    calls = {"templates": 0, "batch_sizes": []}

    def runner(t, s):     # names carry their depth: P<d>_<hash>; deeper reactants are purchasable more often
        calls["templates"] += 1
        d = int(s[1]) if s[0] == "P" else 0
        out = []
        for salt in ("a", "b"):
            h = zlib.crc32((t + s + salt).encode())
            out.append(f"B{h % 50}" if (h >> 8) % 5 < d + 1 else f"P{d + 1}_{h % 9973}")
        return [".".join(out)]
    pred.template_runner = runner
    orch, tok = e2e.build_orchestrator(llm, types.SimpleNamespace(text_input_size=768, check_valid=lambda s: True), dev)
    orch.graph_predictor, orch.graph_encoder = pred, enc
    orch.graph_to_lm_connector = torch.nn.Sequential(torch.nn.Linear(H, llm.config.hidden_size), torch.nn.SiLU()).to(dev, torch.bfloat16)
    x, ei, ea, batch = synth.make_mol_graphs(64, 0, min_atoms=32, max_atoms=32)
    pool = GraphBatch(x, ei, ea, batch, [32] * 64).to_data_list()
    orch.smiles_to_graph = lambda s: type(pool[0])(*(t.clone() for t in (lambda g: (g.x, g.edge_index, g.edge_attr))(pool[zlib.crc32(s.encode()) % 64])))
    orch.enable_mi355x_decode()
    inner = orch.one_step_reaction_batch

    def counted(reqs, topk, **kw):
        calls["batch_sizes"].append(len(reqs))
        return inner(reqs, topk, **kw)
    orch.one_step_reaction_batch = counted
    try:
        kw = dict(expansion_topk=50, iterations=5, starting_mols={f"B{i}" for i in range(50)}, max_planning_time=1e9, rollback=False,
                  design_text="Design", do_sample=True, temperature=0.6, top_p=0.9, top_k=50, max_new_tokens=12,
                  eos_token_id=[], pad_token_id=tok.pad_token_id)
        orch.retro_max_new_tokens = 12
        targets = [f"P0_{i}" for i in range(16)]
        torch.manual_seed(0)
        out = orch.retrosynthesize_many([None] * 16, targets, **kw)
    finally:
        restore_elementwise(llm)
    assert len(out) == 16 and [o["target"] for o in out] == targets
    assert 1 <= len(calls["batch_sizes"]) <= 5 and calls["batch_sizes"][0] == 16 and max(calls["batch_sizes"]) == 16    # lock step, depth <= 5
    assert calls["templates"] >= 16 * 50                                   # every first-round expansion applied its top-50 templates
    n_ok = 0
    for o in out:
        assert set(o) == {"target", "success", "time", "reaction_list", "cost", "templates", "analysis_tokens", "route_length"}
        if o["success"]:
            n_ok += 1
            assert 1 <= o["route_length"] <= 5 and len(o["reaction_list"]) == len(o["templates"]) == len(o["cost"])
            assert o["reaction_list"][0].startswith(o["target"] + ">>") and all(t.startswith("T") for t in o["templates"])
            assert all(0.0 < c <= 1.0 for c in o["cost"])
            made = {o["target"]}                                           # every reaction consumes a molecule an earlier one produced
            for r in o["reaction_list"]:
                p, rs = r.split(">>")
                assert p in made
                made |= set(rs.split("."))
    assert n_ok >= 1


def test_value_estimates_shared_opening_vs_whole_prompts_bf16():
    """The A* value prompts all open with the same tokens; their keys / values are computed once per call and every row's forward covers
    only its remainder (modeling_llamole.estimate_synthesis_complexity_batch).  bf16 on the device, the intended expectation (where the
    logits matter): shared-opening == whole prompts == one forward per node (the reference's structure, modeling_llamole.py:891-993)
    within bf16 rounding of the logits; stated tolerance 2e-2 on costs in [0, 7]."""
    import numpy as np
    from llamole_amd import e2e
    from llamole_amd.planner import ReactionView
    dev = torch.device("cuda")
    llm = e2e.build_llm("tiny", dev, torch.bfloat16)
    orch, tok = e2e.build_orchestrator(llm, types.SimpleNamespace(text_input_size=768, check_valid=lambda s: True), dev)
    orch.expected_cost_value = True
    items = []
    for i in range(70):
        smi = "C" * (1 + i % 17) + "N" * (i % 5) + f"c{i}"
        items.append((smi, None if i % 7 == 0 else ReactionView(1 + i % 4, f"[C:{i}]>>[C:{i}]O" * (1 + i % 3), ["CC" * (1 + i % 6), f"N{i}"])))
    single = [orch.estimate_synthesis_complexity(s, None, r, 0, 1) for s, r in items]
    assert max(single) - min(single) > 1e-3
    shared = orch.estimate_synthesis_complexity_batch(items, None, 0, 1, max_batch=32)
    assert orch.last_value_opening >= 8
    orch.value_prefix_min = 0
    whole = orch.estimate_synthesis_complexity_batch(items, None, 0, 1, max_batch=32)
    assert orch.last_value_opening == 0
    d_sw = float(np.abs(np.array(shared) - np.array(whole)).max())
    d_s1 = float(np.abs(np.array(shared) - np.array(single)).max())
    d_w1 = float(np.abs(np.array(whole) - np.array(single)).max())
    print(f"value estimates, bf16 tiny Qwen2: |shared - whole| {d_sw:.2e}, |shared - single| {d_s1:.2e}, |whole - single| {d_w1:.2e}")
    assert max(d_sw, d_s1, d_w1) < 2e-2
