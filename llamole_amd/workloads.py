"""Synthetic workloads of BASELINE.json configs[2] (retrosynthesis planning) and configs[4] (SFT), for ``bench.py``.

There is no network on the GPU box and no rdkit / rdchiral in the image, so -- exactly as for the e2e workload
(``llamole_amd/e2e.py``) -- the LLM is the named ARCHITECTURE with seeded random-init weights, graphs and conditions are
seeded synthetic tensors of the reference's shapes, and the host chemistry of an expansion (template application, SMILES ->
graph) is scripted: a product maps to one of 64 seeded 32-atom graphs and a (template, product) pair to a two-reactant string.
What is timed is therefore the device side of the reference's loop plus its host logic (A* tree, prompts, batching):

  retro : reference ``GraphLLMForCausalMLM.retrosynthesize`` / ``one_step_reaction`` / ``estimate_synthesis_complexity``
          (src/model/modeling_llamole.py:815-1093) and ``planner/molstar.py`` for a batch of target molecules, preceded by the
          design phase of the same prompts (LLM decode -> query forward -> GraphDiT; :584-663).
  sft   : reference ``GraphLLMForCausalMLM.forward`` (:299-437) under an optimizer step, LoRA on every projection Linear.
"""
from __future__ import annotations

import time
import types
import re
import zlib

import torch

from . import synth


def device_gin_weights(shapes, device, seed: int):
    """Random-init GIN weights of the given shapes drawn ON the device (seeded): the 740 MB template head takes a minute of
    host RNG otherwise.  Same families as synth.make_gin_weights (unit LayerNorm gains, small biases, xavier matrices)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, shp in shapes.items():
        if k.endswith("eps"):
            sd[k] = torch.zeros(1, device=device)
        elif len(shp) == 1:
            gain = k.endswith((".1.weight", "norm1.weight")) or ".norms." in "." + k and k.endswith("weight")
            sd[k] = (1.0 if gain else 0.0) + 0.05 * torch.randn(shp, generator=g, device=device)
        elif "encoder.weight" in k or "embedding" in k or "text_dropping" in k:
            sd[k] = 0.5 * torch.randn(shp, generator=g, device=device)
        else:
            sd[k] = (2.0 / (shp[0] + shp[1])) ** 0.5 * torch.randn(shp, generator=g, device=device)
    return sd


def build_gin_pair(device, out_dim: int, layers: int = 5, hidden: int = 512, templates: bool = True, dtype=torch.bfloat16):
    """GraphCLIP encoder + GraphPredictor at the reference's sizes (graph_encoder / graph_predictor model_config.json: 5 layers,
    hidden 512; USPTO template count 180 576), bf16 parameters = bf16 engines."""
    from .graph_encoder import GraphCLIP
    from .graph_predictor import GraphPredictor
    enc = GraphCLIP(layers, hidden, 0.0, {})
    enc.to(device)
    enc.molecule_encoder.load_state_dict(device_gin_weights(synth.gin_weight_shapes(layers, hidden, "encoder"), device, 1))
    enc.molecule_projection.load_state_dict(device_gin_weights(synth.proj_weight_shapes(hidden), device, 2))
    pred = GraphPredictor(layers, hidden, 0.0, out_dim, {}, {i: f"T{i}" for i in range(out_dim)} if templates else {})
    pred.to(device)
    sd_pred = device_gin_weights(synth.gin_weight_shapes(layers, hidden, "predictor", out_dim), device, 3)
    pred.predictor.load_state_dict(sd_pred)
    for m in (enc, pred):
        for p in m.parameters():
            p.data = p.data.to(dtype)
            p.requires_grad = False
    return enc, pred, sd_pred


# ------------------------------------------------------------------------------------------------------------ retro (configs[2])
def build_retro_step(args, graph_decoder, device, rank: int, world: int = 1):
    """step_fn(i) -> (designed molecule graphs, per-target records [targets, 3] f32 = (succeeded, route length, route cost)).  One step = the design phase for `targets` prompts as ONE batch (LLM decode of the analysis, query forward,
    GraphDiT reverse diffusion) + `targets` A* searches run in lock step (reference: one after the other, :1173-1190) with at most
    `iterations` expansions each (search depth <= iterations)."""
    from . import e2e
    from .graph_data import GraphBatch
    llm = e2e.build_llm(args.llm, device, torch.bfloat16)
    enc, pred, sd_pred = build_gin_pair(device, args.out_dim)
    # Scripted chemistry (rdkit / rdchiral are in neither image) with a purchasable set, so that the searches PLAN something: molecule names
    # carry their fate.  `S<D>d<d>_*` lies on a route of D reactions at depth d: EVERY template turns it into the same pair -- one purchasable
    # building block `B*` plus the next intermediate, or two building blocks at depth D - 1 -- which sample_templates merges into one
    # candidate (reactant sets are de-duplicated, graph_predictor/model.py:190-228), so the search closes after exactly D expansions
    # (early exit, route extraction and reaction-list assembly all run inside the timed region).  `U_d<d>_*` decomposes into a different
    # pair of non-purchasable molecules under every template, for ever: that search spends its whole expansion budget on a widening tree
    # and fails, like the reference's 30 s / 100 iteration budget running dry (eval/workflow.py:171-173).  Half of a step's targets are of
    # each kind; D cycles through 2, 3, 4.
    def template_runner(t, s):
        if s[0] == "S":
            h = zlib.crc32(s.encode())
            D, d = (int(v) for v in re.match(r"S(\d+)d(\d+)_", s).groups())
            if d + 1 >= D:
                return [f"B{h % 50}.B{(h >> 8) % 50}"]
            return [f"B{h % 50}.S{D}d{d + 1}_{h % 9973}"]
        h = zlib.crc32((t + s).encode())
        d = int(re.match(r"U_d(\d+)_", s).group(1)) if s[0] == "U" else 0
        return [f"U_d{d + 1}_{h % 9973}.U_d{d + 1}_{(h >> 8) % 9973}"]
    pred.template_runner = template_runner
    purchasable = {f"B{i}" for i in range(50)}
    # the orchestrator only asks its graph decoder for the condition width and for SMILES validity (rdkit: scripted here); the reverse
    # diffusion itself is called on the real engine below
    orch, tok = e2e.build_orchestrator(llm, types.SimpleNamespace(text_input_size=768, check_valid=lambda s: True), device)
    orch.graph_predictor, orch.graph_encoder = pred, enc
    orch.graph_to_lm_connector = torch.nn.Sequential(torch.nn.Linear(enc.hidden_size, llm.config.hidden_size), torch.nn.SiLU()).to(device, torch.bfloat16)
    x, ei, ea, batch = synth.make_mol_graphs(64, 0, min_atoms=32, max_atoms=32)
    pool = GraphBatch(x, ei, ea, batch, [32] * 64).to_data_list()
    orch.smiles_to_graph = lambda s: type(pool[0])(*(t.clone() for t in (lambda g: (g.x, g.edge_index, g.edge_attr))(pool[zlib.crc32(s.encode()) % 64])))
    accel = orch.enable_mi355x_decode()
    orch.constant_language_cost_shortcut = bool(getattr(args, "retro_constant_value", False))      # opt-in; default: every value forward runs
    # --total-targets N (strong scaling): the N searches of a step are ONE lock-step problem for all ranks -- every rank runs the same
    # host A*, each round's expansions and value prompts are split over the ranks (expansion_shard: one all-gather of top-k records +
    # one of costs per round); the design phase of the step is split by prompt.  Default (weak scaling): `targets` searches per GPU.
    total = int(getattr(args, "total_targets", 0) or 0)
    T = total if total else args.targets
    Td = len(range(rank, T, world)) if total else T          # prompts this rank designs
    if total and world > 1:
        orch.expansion_shard = (rank, world, None)
    kw = dict(expansion_topk=args.topk, iterations=args.iterations, starting_mols=purchasable, max_planning_time=1e9, rollback=False,
              design_text="Design", do_sample=True, temperature=0.6, top_p=0.9, max_new_tokens=args.retro_tokens,
              eos_token_id=[], pad_token_id=tok.pad_token_id)
    orch.retro_max_new_tokens = args.retro_tokens
    g = torch.Generator().manual_seed(100 + rank)
    prompt = torch.randint(5, 1000, (max(Td, 1), args.cutoff_len), generator=g).to(device)
    mask = torch.ones_like(prompt)
    dkw = e2e.gen_kwargs(tok, args.new_tokens)
    props, _, _ = synth.make_dit_inputs(max(Td, 1), seed=rank, max_node=graph_decoder.max_n_nodes)
    n_nodes = torch.full((max(Td, 1),), graph_decoder.max_n_nodes, dtype=torch.int64)
    last = {}
    count = {"expansions": 0, "value_estimates": 0, "value_calls": 0}
    expand, values = orch.one_step_reaction_batch, orch.estimate_synthesis_complexity_batch

    spans = []      # (start, end) HIP events around every value-estimate call: their GPU time is read after the step's synchronisation

    def counted_expand(reqs, *a, **k):
        count["expansions"] += len(reqs)
        return expand(reqs, *a, **k)

    def counted_values(items, *a, **k):
        count["value_estimates"] += len(items)
        count["value_calls"] += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = values(items, *a, **k)
        e1.record()
        spans.append((e0, e1))
        return out
    orch.one_step_reaction_batch, orch.estimate_synthesis_complexity_batch = counted_expand, counted_values

    def step_fn(i):
        torch.manual_seed(1000 * rank + i)
        t0 = time.perf_counter()
        _, _, cond = orch.design_hidden(prompt, mask, None, **dkw)
        mols, _ = graph_decoder.generate_graphs(props, cond.float(), -200.0, n_nodes=n_nodes, seed=1000 * rank + i)
        t1 = time.perf_counter()
        tr = 0 if total else rank                                   # strong scaling: the same targets on every rank
        targets = [(f"S{2 + (j // 2) % 3}d0_{tr}_{i}_{j}" if j % 2 == 0 else f"U_d0_{tr}_{i}_{j}") for j in range(T)]
        orch.value_tokens_forwarded = 0
        routes = orch.retrosynthesize_many([None] * T, targets, **kw)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rec = torch.zeros(T, 3)
        for j, r in enumerate(routes):
            cost = r["cost"]           # exp(-cost) per reaction of the extracted route (SynRoute.get_reaction_list): the record keeps their product
            prob = float(torch.tensor(cost, dtype=torch.float64).prod()) if isinstance(cost, (list, tuple)) and cost else float(cost or 0.0)
            rec[j] = torch.tensor([float(bool(r["success"])), float(r["route_length"] or 0), prob])
        value_s = sum(a.elapsed_time(b) for a, b in spans) * 1e-3
        del spans[:]
        last.update(design_s=t1 - t0, retro_s=t2 - t1, value_forward_s=value_s, value_tokens=getattr(orch, "value_tokens_forwarded", 0),
                    value_prompt_opening_tokens=getattr(orch, "last_value_opening", 0))
        return mols, rec

    step_fn.count = count

    info = {"llm": args.llm, "llm_weights": "random-init (no network)", "targets_per_gpu": T, "max_expansions_per_search": args.iterations,
            "analysis_tokens_per_expansion": args.retro_tokens, "expansion_topk": args.topk, "templates": args.out_dim,
            "chemistry": "scripted (rdkit / rdchiral are not in this image): product -> one of 64 seeded 32-atom graphs, "
                         "(template, product) -> a two-reactant string; 50 purchasable building blocks; every second target has a route of "
                         "2 / 3 / 4 reactions (the search exits early and its route is extracted inside the timed region), the others have none "
                         "and spend the whole expansion budget",
            "search": "lock-step A* over the batch: batched GIN encode / LLM decode / predictor + top-k / value forward per expansion round "
                      "(reference: searches one after the other, one LLM forward per new tree node)",
            "value_estimates": ("constant-cost shortcut (opt-in): the reference-compatible language cost is 15 for every molecule, returned without the "
                                "LLM forward" if orch.constant_language_cost_shortcut else
                                f"one left-padded LLM prefill per {orch.value_batch} new tree nodes, every forward executed (the reference: one forward per node); "
                                "the tokens every prompt opens with (timing_breakdown.value_prompt_opening_tokens) are forwarded once per call and "
                                "enter every row as cached keys / values"),
            "llm_acceleration": accel, "timing_breakdown": last}
    return step_fn, info, orch, llm, sd_pred


# ------------------------------------------------------------------------------------------------------------ SFT (configs[4])
def build_sft_step(args, device, rank: int):
    """step_fn(i) -> log dict of one optimizer step on this rank's batch (the gradient all-reduce of `sft_step` is the path's only
    collective).  Batch: `sft_batch` rows of `sft_seq` tokens, one spliced molecule and two retro queries per row."""
    from . import e2e
    from .graph_data import GraphBatch
    from .modeling_llamole import SPECIAL_TOKENS, GraphLLMForCausalMLM
    from .sft import GraphSFTCollator, add_lora, sft_step, to_device
    llm = e2e.build_llm(args.llm, device, torch.bfloat16)
    n_lora = add_lora(llm)
    enc, pred, sd_pred = build_gin_pair(device, args.out_dim, templates=False)
    V = llm.config.vocab_size
    tid = {t: V - 19 + i for i, t in enumerate(SPECIAL_TOKENS)}
    model = GraphLLMForCausalMLM(types.SimpleNamespace(), types.SimpleNamespace(loss_weight_lm=1, loss_weight_design=1, loss_weight_retro=1),
                                 types.SimpleNamespace(learned_query_size=8), llm, types.SimpleNamespace(text_input_size=768), pred, enc, tid, None)
    torch.manual_seed(1)
    for nm in ("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor"):
        getattr(model, nm).to(device=device, dtype=torch.bfloat16)
    x, ei, ea, batch = synth.make_mol_graphs(8, 0, min_atoms=32, max_atoms=32)
    graphs = dict(enumerate(GraphBatch(x, ei, ea, batch, [32] * 8).to_data_list()))
    g = torch.Generator().manual_seed(rank)
    B, S = args.sft_batch, args.sft_seq
    feats = []
    for i in range(B):
        ids = torch.randint(5, V - 64, (S,), generator=g).tolist()
        ids[7] = tid["<molecule>"]
        for start in (S // 3, 2 * S // 3):
            ids[start] = tid["<retro_start>"]
            ids[start + 1:start + 9] = [tid["<retro_body>"]] * 8
        feats.append({"input_ids": ids, "labels": [-100] * 16 + ids[16:], "molecule_ids": [i % 8],
                      "retro_product_ids": [(i + 1) % 8, (i + 2) % 8],
                      "retro_labels": [int(torch.randint(0, args.out_dim, (1,), generator=g)) for _ in range(2)]})
    b = to_device(GraphSFTCollator(0, graphs)(feats), device)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4)
    last = {}

    def step_fn(i):
        log = sft_step(model, b, opt)
        last.update(log)
        return log

    def graph_side_ms(iters: int = 10):
        """The graph side of the step alone: encoder forward + predictor forward / cross-entropy / reverse sweep on the same batch."""
        rp, mg = b["retro_product_graphs"], b["molecule_graphs"]
        c = torch.randn(rp.num_graphs, 768, device=device, dtype=torch.bfloat16)
        lab = b["retro_labels"].flatten()

        def run():
            enc(mg.x, mg.edge_index, mg.edge_attr, mg.batch)
            cg = c.clone().requires_grad_(True)
            torch.nn.functional.cross_entropy(pred(rp.x, rp.edge_index, rp.edge_attr, rp.batch, cg).float(), lab).backward()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / iters

    step_fn.graph_side_ms = graph_side_ms
    info = {"llm": args.llm, "llm_weights": "random-init (no network)", "lora": f"r=8, alpha=16 on {n_lora} projection Linears (llamole_amd.sft.add_lora; "
            "peft is not in this image)", "trainable_params": sum(p.numel() for p in params), "rows_per_gpu": B, "tokens_per_row": S,
            "per_row": "1 spliced molecule graph (GIN encoder) + 2 retro queries (GIN predictor forward + reverse sweep w.r.t. the condition)",
            "gin": {"hidden": 512, "layers": 5, "templates": args.out_dim}, "optimizer": "AdamW (torch), bf16 parameters",
            "llm_fwd_bwd": "stock HuggingFace on PyTorch-ROCm autograd (hipBLASLt); graph side of the loss in libllamole_hip", "last_log": last}
    return step_fn, info, model, sd_pred, b
