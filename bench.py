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

from llamole_amd import synth  # noqa: E402

LLM_LABEL = {"qwen2-7b": "Qwen2-7B", "llama-3.1-8b": "Llama-3.1-8B", "mistral-7b": "Mistral-7B"}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0


def dit_step_bytes(H, L, Hm, N, B, esz):
    """Algorithmic HBM bytes of one reverse step: in-loop weights once
    (qkv, proj, fc1, fc2 per block + decoder; adaLN weights are hoisted out of the loop) plus the
    hoisted modulation rows, the int8 state and the decoder output."""
    F = 16 + 5 * N
    w = L * (3 * H * H + H * H + 2 * H * Hm) + H * H + F * H
    mod = (B + 1) * (L * 6 * H + 2 * F) * 4
    state = 2 * (B * N + B * N * N) + 2 * B * N * F * 4
    return w * esz + mod + state


def dit_step_flops(H, L, Hm, N, B):
    F = 16 + 5 * N
    M2 = 2 * B * N
    lin = 2 * M2 * (L * (4 * H * H + 2 * H * Hm) + H * H + F * H)
    att = 2 * B * L * 4 * N * N * H
    return lin + att


def log(*a):
    if os.environ.get("BENCH_VERBOSE"):
        print("[bench %.1fs]" % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def fast_dit_weights(cfg, max_node, device):
    """Random-init weights of the reference denoiser's shapes, drawn on the device (seeded):
    same distributions as synth.make_dit_weights, without the minute of host RNG at 573 M params."""
    g = torch.Generator(device=device).manual_seed(1234)
    sd = {}
    for k, shp in synth.dit_weight_shapes(cfg, max_node).items():
        if len(shp) == 1:
            gain = k.endswith(("norm.weight", "x_embedder.1.weight"))
            sd[k] = (1.0 if gain else 0.0) + (0.1 if gain else 0.05) * torch.randn(shp, generator=g, device=device)
        elif "embedding" in k:
            sd[k] = 0.5 * torch.randn(shp, generator=g, device=device)
        else:
            sd[k] = (2.0 / (shp[0] + shp[1])) ** 0.5 * torch.randn(shp, generator=g, device=device)
    return sd


def build_model(args, device):
    import tempfile
    from llamole_amd.graph_decoder import GraphDiT
    cfg = synth.make_dit_config(args.hidden, args.depth, args.heads, args.T, args.guide)
    meta = synth.make_data_meta(args.nodes, 0, fixed_n_nodes=args.nodes)
    sd = fast_dit_weights(cfg, args.nodes, device)
    log("weights drawn")
    d = tempfile.mkdtemp()
    synth.write_dit_dir(d, cfg, meta, {})          # config.yaml + data.meta.json only
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    m = GraphDiT(os.path.join(d, "config.yaml"), os.path.join(d, "data.meta.json"), dtype)
    m.to(device)
    m.denoiser.load_state_dict(sd)
    if dtype != torch.float32:
        for p in m.parameters():
            p.data = p.data.to(dtype)
    return m, cfg, meta, sd


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but grants a 16-CPU quota; oversubscribing it stalls)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("BENCH_CPU_THREADS", "64"))))


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
            "sample": f"HF {args.llm} (random-init, bf16) on {cores} CPU threads: prompt pass {prefill:.2f} s, "
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


def time_dominant_kernel(args, device):
    """The dominant hand-written kernel of the workload, timed by HIP events on the stream it is launched on, back to
    back over enough distinct weight matrices to defeat the 256 MiB Infinity Cache (ll_gemm_bench in the C ABI):
      e2e      : the weight-streaming kernel under the LLM's gated-MLP gate|up projection at decode (15.2 GB of bf16 weights
                 per token for Qwen2-7B), shape [2 x 18944 x 3584], M = batch: gemv_fused_kernel (M <= 2) or rows16_kernel (3..16);
      graphdit : gemm_bf16_pipe_kernel at the block-MLP fc1 shape, M = 2*B*N tokens.
    Returns (avg_ms, algorithmic bytes, flops, name, pmc key)."""
    import ctypes as C
    from llamole_amd import _lib
    lib = _lib.load()
    if args.dtype != "bf16":
        return None
    if args.workload == "e2e" and args.llm_linear == "hip":
        from llamole_amd.e2e import LLM_CONFIGS
        spec = LLM_CONFIGS[args.llm]
        M, N, K = args.batch, spec["intermediate_size"], spec["hidden_size"]
        ms = C.c_float()
        if args.llm_fuse and args.llm_layer_fuse and args.llm_decode != "hf" and 3 <= M <= 16 and K % 32 == 0:
            # batched decode: the same projection on the weight-streaming MFMA Linear (ll_linear_rows16_bf16)
            rows = 2 * N
            nw = max(2, int(600e6 // (rows * K * 2)))
            _lib.check(lib.ll_rows16_bench(M, N, K, 2, 1, 8 * nw, nw, C.byref(ms)), "ll_rows16_bench")
            name = (f"rows16_kernel<silu_mul,256,norm>, LLM gated-MLP gate|up projection [{M}x{K}]x[{rows}x{K}]^T bf16 + RMSNorm prologue "
                    f"+ SiLU*mul epilogue (batched decode step)")
            nbytes = rows * K * 2 + M * K * 2 + K * 2 + M * N * 2
            return ms.value, nbytes, 2.0 * M * rows * K, name, f"llm_rows16_m{M}_n{rows}_k{K}"
        if args.llm_fuse and args.llm_layer_fuse and args.llm_decode != "hf" and M <= 2:
            # the fused layer's gated-MLP kernel: RMSNorm prologue, gate|up rows streamed once, SiLU*mul epilogue
            rows = 2 * N
            nw = max(2, int(600e6 // (rows * K * 2)))
            _lib.check(lib.ll_gemv_fused_bench(M, N, K, 2, 1, 1, 8 * nw, nw, C.byref(ms)), "ll_gemv_fused_bench")
            name = (f"gemv_fused_kernel<{M},norm,silu_mul,nt>, LLM gated-MLP gate|up projection "
                    f"[{M}x{K}]x[{rows}x{K}]^T bf16 + RMSNorm prologue + SiLU*mul epilogue (decode step)")
            nbytes = rows * K * 2 + M * K * 2 + K * 2 + M * N * 2
            return ms.value, nbytes, 2.0 * M * rows * K, name, f"llm_gemv_fused_m{M}_n{rows}_k{K}"
        name = f"gemv_bf16_kernel, LLM MLP up-projection [{M}x{K}]x[{N}x{K}]^T bf16 (decode step)"
        key = f"llm_gemv_m{M}_n{N}_k{K}"
    else:
        return time_graphdit_kernel(args, args.batch)
    nw = max(2, int(600e6 // (N * K * 2)))
    ms = C.c_float()
    _lib.check(lib.ll_gemm_bench(M, N, K, -1, 1, 0, 4 * nw, nw, C.byref(ms)), "ll_gemm_bench")
    nbytes = N * K * 2 + M * K * 2 + M * N * 2
    flops = 2.0 * M * N * K
    return ms.value, nbytes, flops, name, key


def time_graphdit_kernel(args, batch: int):
    """The dominant kernel of the GraphDiT step at `batch` graphs: the block-MLP fc1 GEMM at M = 2 * batch * N token rows, on whatever
    kernel the production dispatch picks for that M (32-33 % of the step's GPU time in profiles/r*_graphdit_b{1,8}_kernel_stats.csv),
    timed by HIP events over distinct weight matrices (ll_gemm_bench).  Returns (avg_ms, algorithmic bytes, flops, name, pmc key)."""
    import ctypes as C
    from llamole_amd import _lib
    lib = _lib.load()
    H, Hm = args.hidden, int(args.hidden * 4)
    M, N, K = 2 * batch * args.nodes, Hm, H
    kern = ("gemm_m64_kernel<8,8,bf16,packed>" if M <= 64 else "gemm_m128_kernel" if M <= 224 else
            "gemm_bf16_pipeu_kernel<64,64,4,4,4>" if M < 1024 else
            "gemm_bf16_pipe_kernel<128,128,4,4,3>" if M < 2048 else "gemm_bf16_pipe_kernel<256,128,4,4,3>")
    name = f"{kern}, GraphDiT block-MLP fc1 [{M}x{K}]x[{N}x{K}]^T bf16"
    key = f"fc1_m{M}"
    nw = max(2, int(600e6 // (N * K * 2)))
    ms = C.c_float()
    _lib.check(lib.ll_gemm_bench(M, N, K, -1, 1, 0, 4 * nw, nw, C.byref(ms)), "ll_gemm_bench")
    nbytes = N * K * 2 + M * K * 2 + M * N * 2
    flops = 2.0 * M * N * K
    return ms.value, nbytes, flops, name, key


def graphdit_kernel_profile_avg(args, batch: int):
    """(avg us, launches, kernel name, file) of the fc1 GEMM instantiation inside the GraphDiT step at `batch` graphs, read from the committed
    rocprofv3 kernel-trace summary of `bench.py --workload graphdit --batch <batch>` (profiles/r4_graphdit_b<batch>_step_kernel_stats.csv), or
    None when there is no trace of this shape / denoiser."""
    import csv
    if (args.hidden, args.depth, args.nodes, args.dtype) != (1024, 28, 32, "bf16"):
        return None
    M = 2 * batch * args.nodes
    want = ("gemm_m64_kernel<8, 8, unsigned short, true>" if M <= 64 else
            "gemm_bf16_pipeu_kernel<64, 64, 4, 4, 4, unsigned short>" if 224 < M < 1024 else None)
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{r}_graphdit_b{batch}_step_kernel_stats.csv") for r in (5, 4)) if os.path.exists(q)), None)
    if want is None or path is None:
        return None
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["kernel"] == want:
                return float(row["avg_us"]), int(row["calls"]), want, os.path.relpath(path, ROOT)
    return None


def time_kernel_class_in_situ(m, cls: str, props, text, n_nodes, mode: str = "bracket"):
    """(mean ms per pair, pairs) for one class of the block's kernels inside ONE launched trajectory of the engine `m` (tuning hook
    ll_dit_class_probe: HIP events on the stream the trajectory runs on).  mode "bracket": the pair brackets every launch of the class;
    "empty": the pair is recorded back to back at the same launch site -- what an event pair itself adds there."""
    import ctypes as C
    from llamole_amd import _lib
    lib = _lib.load()
    classes = {"qkv": 0, "attn": 1, "proj": 2, "lnmod": 3, "fc1": 4, "fc2": 5}
    try:
        _lib.check(lib.ll_dit_class_probe(m._handle, classes[cls] | (0x100 if mode == "empty" else 0)), "ll_dit_class_probe")
        m.generate_graphs(props, text, -200.0, n_nodes=n_nodes, seed=777, use_graph=False)
        us, n = C.c_float(), C.c_int()
        _lib.check(lib.ll_dit_class_probe_read(m._handle, C.byref(us), C.byref(n)), "ll_dit_class_probe_read")
        _lib.check(lib.ll_dit_class_probe(m._handle, -1), "ll_dit_class_probe")
    except Exception as e:      # noqa: BLE001
        log("in-situ kernel timing failed:", e)
        return None
    if n.value == 0:
        return None
    return us.value / n.value * 1e-3, n.value


def time_fc1_marginal(m, props, text, n_nodes, depth: int, T: int, reps: int = 3):
    """ms that ONE fc1 launch adds to the launched trajectory: (trajectory with fc1) - (the same trajectory with the fc1 launches left out,
    ll_dit_class_probe SKIP: timing only, its molecules are garbage), over depth x T launches; best of `reps` each.  The kernel plus its
    share of the launch boundary, as the dependent chain pays for it."""
    from llamole_amd import _lib
    lib = _lib.load()
    try:

        def traj(skip):
            _lib.check(lib.ll_dit_class_probe(m._handle, (4 | 0x200) if skip else -1), "ll_dit_class_probe")
            best = float("inf")
            for _ in range(reps):
                m.generate_graphs(props, text, -200.0, n_nodes=n_nodes, seed=778, use_graph=False)
                best = min(best, m.last_run_ms()[0])
            return best
        with_fc1 = traj(False)
        without = traj(True)
        _lib.check(lib.ll_dit_class_probe(m._handle, -1), "ll_dit_class_probe")
    except Exception as e:      # noqa: BLE001
        log("marginal fc1 timing failed:", e)
        return None
    return (with_fc1 - without) / (depth * T), with_fc1, without


def roofline_object(args, dom):
    """`roofline` of the JSON line from (avg_ms, algorithmic bytes, flops, kernel name, pmc key): HBM- or MFMA-bound by which peak the
    kernel's algorithmic work would take longer on; `traffic` from the PMC passes committed under profiles/ (same kernel, same shape)."""
    kms, kbytes, kflops, kname, kkey = dom
    hbm_t, mfma_t = kbytes / (HBM_PEAK_GBS * 1e9), kflops / (MFMA_BF16_PEAK_TF * 1e12)
    if hbm_t >= mfma_t or args.dtype != "bf16":
        roof = {"bound": "hbm", "achieved": kbytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    else:
        roof = {"bound": "mfma", "achieved": kflops / (kms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"] = None
    for pmc_file in ("r3_pmc_traffic.json", "r2_pmc_traffic.json"):     # r1/r2 entries: LLM kernels (frozen since) and the GraphDiT fc1 GEMMs
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
        except Exception:
            continue
        if kkey in pmc and args.dtype == "bf16" and (not kkey.startswith("fc1") or args.hidden == 1024):
            roof["traffic"] = pmc[kkey]["hbm_bytes_per_launch"]
            roof["traffic_source"] = f"profiles/{pmc_file} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, FETCH doubled per MI355X_MICROARCH.md)"
            break
    roof["algorithmic_bytes"] = kbytes
    roof["algorithmic_flops"] = kflops
    roof["kernel"] = kname
    roof["kernel_ms"] = kms
    return roof


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



def _barrier(ctx):
    if ctx.dist is not None:
        ctx.dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(ctx, dt: float) -> float:
    if ctx.dist is None:
        return dt
    t = torch.tensor([dt], device=ctx.device if ctx.dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    ctx.dist.all_reduce(t, op=ctx.dist.ReduceOp.MAX)
    return float(t.item())


def time_template_head(args, graphs: int):
    """The weight stream of the GIN predictor's template head ([graphs, 4H] x [out_dim, 4H]^T, 740 MB of bf16 at 180 576 templates:
    rows16_kernel<plain, f32 out>), timed by HIP events on its own stream over two distinct weight copies (ll_rows16_bench)."""
    import ctypes as C
    from llamole_amd import _lib
    lib = _lib.load()
    K, N, M = 4 * 512, args.out_dim, max(3, min(16, graphs))
    ms = C.c_float()
    _lib.check(lib.ll_rows16_bench(M, N, K, 0x100, 0, 16, 2, C.byref(ms)), "ll_rows16_bench")
    nbytes = N * K * 2 + M * K * 2 + M * N * 4 + N * 4
    roof = {"bound": "hbm", "achieved": nbytes / (ms.value * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"] = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")))
        key = f"gin_head_rows16_m{M}_n{N}_k{K}"
        if key in pmc:
            roof["traffic"] = pmc[key]["hbm_bytes_per_launch"]
            roof["traffic_source"] = "profiles/r3_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, FETCH doubled per MI355X_MICROARCH.md)"
    except Exception:
        pass
    roof["algorithmic_bytes"] = nbytes
    roof["kernel"] = (f"rows16_kernel<plain, f32 out>, GIN predictor template head [{M}x{K}]x[{N}x{K}]^T bf16 (decoder.4 of "
                      f"graph_predictor/model.py:272-278)")
    roof["kernel_ms"] = ms.value
    return roof


def value_forward_mfma(llm, tokens: int, seconds: float):
    """The A* value forwards of the timed steps against the dense bf16 MFMA peak: 2 x (decoder-stack parameters) x (tokens forwarded) flops
    over the HIP-event time of the calls (host tokenisation between their launches included) -- the step's dominant cost, vendor GEMMs
    under the stock HF forward, reported next to the hand-written kernel's `roofline`."""
    if not tokens or seconds <= 0:
        return None
    params = sum(p.numel() for n, p in llm.named_parameters() if "embed_tokens" not in n and "lm_head" not in n)
    tf = 2.0 * params * tokens / seconds / 1e12
    return {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0, "tokens": int(tokens),
            "kernel": "hipBLASLt MT256x256x64 GEMMs under the stock HF prefill (77 % of the forward, profiles/r3_value_forward_kernel_stats.csv)"}


def run_retro(args, ctx):
    """BASELINE.json configs[2]: design + lock-step A* retrosynthesis for `targets` prompts per GPU (llamole_amd/workloads.py)."""
    from llamole_amd.workloads import build_retro_step
    m, cfg, meta, sd = build_model(args, ctx.device)
    step_fn, info, orch, llm, sd_pred = build_retro_step(args, m, ctx.device, ctx.rank, ctx.world)
    log("retro workload built")
    for i in range(args.warmup):
        step_fn(i)
    _barrier(ctx)
    step_fn.count.update(expansions=0, value_estimates=0, value_calls=0)
    t0 = time.perf_counter()
    recs, design_s, retro_s, value_s, value_tokens = [], 0.0, 0.0, 0.0, 0
    for i in range(args.steps):
        _maybe_fail(ctx.rank, i)
        mols, rec = step_fn(args.warmup + i)
        recs.append(rec)
        design_s += info["timing_breakdown"]["design_s"]
        retro_s += info["timing_breakdown"]["retro_s"]
        value_s += info["timing_breakdown"]["value_forward_s"]
        value_tokens += info["timing_breakdown"].get("value_tokens", 0)
    _barrier(ctx)
    dt = _max_over_ranks(ctx, time.perf_counter() - t0)
    n_exp, n_val = step_fn.count["expansions"], step_fn.count["value_estimates"]
    gathered = recs[-1]
    if ctx.dist is not None:
        # the path's only exchange: one all-gather of fixed-size per-target route records (and of the expansion counts)
        dev = ctx.device if ctx.dist.get_backend() == "nccl" else "cpu"
        bufs = [torch.empty_like(recs[-1], device=dev) for _ in range(ctx.world)]
        ctx.dist.all_gather(bufs, recs[-1].to(dev))
        gathered = torch.cat([b.cpu() for b in bufs])
        cnt = torch.tensor([n_exp, n_val], device=dev, dtype=torch.float64)
        ctx.dist.all_reduce(cnt)
        n_exp, n_val = int(cnt[0].item()), int(cnt[1].item())
        if getattr(args, "total_targets", 0):          # replicated A*: every rank counted every expansion / value prompt of the job
            n_exp, n_val = n_exp // ctx.world, n_val // ctx.world
    if ctx.rank != 0:
        return None
    strong = bool(getattr(args, "total_targets", 0))
    T = args.total_targets if strong else args.targets
    if strong:
        gathered = recs[-1]                  # every rank holds all routes already (replicated A*)
    roof = time_template_head(args, T if not strong else max(3, T // ctx.world))
    roof["note"] = ("dominant HAND-WRITTEN kernel of the workload; the step's dominant kernel overall is hipBLASLt's MT256x256x64 GEMM under the stock "
                    "HF forward of the A* value estimates (78 % of that forward, profiles/r3_value_forward_kernel_stats.csv; see value_forward_share_of_step)")
    out = {"metric": "retrosynthesis-planned molecules/sec (design + A* search, depth <= %d)" % args.iterations,
           "value": (T if strong else ctx.world * T) * args.steps / dt, "unit": "molecules/s", "n_gpus": ctx.n_ranks, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "bf16",
           "data": "synthetic",
           "config": {"workload": "%s + GraphDiT + GIN predictor A* retrosynthesis, depth<=%d, %s, %d analysis tokens per expansion (the reference allows 512)"
                                  % (LLM_LABEL.get(args.llm, args.llm), args.iterations,
                                     ("%d searches per step as ONE lock-step problem, expansions / value prompts of every round split over %d GPU(s)" % (T, ctx.n_ranks))
                                     if strong else "batch=%d/GPU" % T, args.retro_tokens),
                      "prompts_per_step": T if strong else ctx.world * T, "gathered_routes": int(gathered.shape[0]),
                      "denoiser": {"hidden": args.hidden, "depth": args.depth, "heads": args.heads, "max_nodes": args.nodes, "T": args.T,
                                   "guide_scale": args.guide}, **{k: v for k, v in info.items() if k != "timing_breakdown"}},
           "expansions_per_s": n_exp / dt, "expansions": n_exp, "value_estimates_per_expansion": n_val / max(1, n_exp),
           "design_share_of_step": design_s / max(1e-9, design_s + retro_s),
           "value_forward_share_of_step": value_s / max(1e-9, design_s + retro_s),
           "value_forward_mfma": value_forward_mfma(llm, value_tokens, value_s),
           "value_prompts_per_call": n_val * 1.0 / max(1, step_fn.count["value_calls"] * ctx.world),
           "value_prompt_opening_tokens": info["timing_breakdown"].get("value_prompt_opening_tokens", 0),
           "value_forward_note": "A* value estimates: the new tree nodes of ALL searches of a round in one call (~100 nodes per expansion, ~140 tokens "
                                 "each), one left-padded LLM prefill per %d prompts, the tokens every prompt opens with forwarded once per call -- stock "
                                 "HF forward on PyTorch-ROCm / hipBLASLt at M ~ 100 k rows, compute-bound (~1.05 PFLOP/s over the decoder stack); the "
                                 "reference runs one forward per node" % orch.value_batch,
           "routes_found": int(gathered[:, 0].sum().item()),
           "route_lengths": sorted(int(v) for v in gathered[gathered[:, 0] > 0, 1].tolist()),
           "searches_without_route": int((gathered[:, 0] == 0).sum().item()),
           "roofline": roof}
    if not args.no_cpu_baseline and ctx.world == 1:
        log("cpu baseline ...")
        out["cpu_baseline"] = cpu_baseline_retro(args, llm, sd_pred, usable_cores(), n_val / max(1, n_exp))
    return out


def sft_llm_mfma(model, rows: int, seq: int, step_s: float):
    """One rank's LLM forward + backward against the dense bf16 MFMA peak: frozen base weights need the forward product and the input-gradient
    product (4 x parameters x tokens flops; no weight-gradient GEMMs, the rank-r LoRA terms are < 1 %), causal attention 4 x rows x heads x
    seq^2 x head_dim / 2 forward and 2.5 x that in the reverse sweep, lm_head forward + input gradient -- over the WHOLE step time."""
    llm = getattr(model, "language_model", model)
    cfg = llm.config
    stack = sum(p.numel() for n, p in llm.named_parameters() if "embed_tokens" not in n and "lm_head" not in n and "lora_" not in n)
    tokens = rows * seq
    head = cfg.vocab_size * cfg.hidden_size
    attn = 4.0 * rows * cfg.num_attention_heads * seq * seq * (cfg.hidden_size // cfg.num_attention_heads) / 2 * cfg.num_hidden_layers
    flops = 4.0 * (stack + head) * tokens + 3.5 * attn
    tf = flops / step_s / 1e12
    return {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0, "flops_per_step": flops,
            "kernel": "hipBLASLt GEMMs + flash attention under the stock HF forward / backward (profiles/r3_sft_kernel_stats.csv)"}


def run_sft(args, ctx):
    """BASELINE.json configs[4]: one SFT optimizer step per bench step, data-parallel over the ranks (llamole_amd/workloads.py)."""
    from llamole_amd.workloads import build_sft_step
    step_fn, info, model, sd_pred, batch = build_sft_step(args, ctx.device, ctx.rank)
    log("sft workload built")
    for i in range(args.warmup):
        step_fn(i)
    _barrier(ctx)
    t0 = time.perf_counter()
    for i in range(args.steps):
        _maybe_fail(ctx.rank, i)
        logd = step_fn(args.warmup + i)
    _barrier(ctx)
    dt = _max_over_ranks(ctx, time.perf_counter() - t0)
    if ctx.rank != 0:
        return None
    B, S = args.sft_batch, args.sft_seq
    n_retro = int(batch["retro_product_graphs"].num_graphs)
    graph_ms = step_fn.graph_side_ms()
    roof = time_template_head(args, n_retro)
    roof["kernel"] += "; the reverse sweep streams the same 740 MB once more (dlogits x W, 16-way split-K)"
    roof["note"] = ("dominant HAND-WRITTEN kernel of the workload (graph side of the loss, graph_side_share of the step); the step itself is the stock HF "
                    "forward / backward of the LLM under PyTorch autograd (hipBLASLt GEMMs)")
    out = {"metric": "SFT samples/sec (LM loss + retro cross-entropy, LoRA)", "value": ctx.world * B * args.steps / dt, "unit": "samples/s",
           "n_gpus": ctx.n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "%s LoRA SFT (LLM fwd/bwd + GIN encoder + GIN predictor fwd/bwd), %d x %d tokens/GPU, data-parallel"
                                  % (LLM_LABEL.get(args.llm, args.llm), B, S),
                      "global_batch": ctx.world * B, "seq_len": S, "parallelism": "dp%d" % ctx.world,
                      "gradient_exchange": "one direct all-reduce per 64 MB bucket of the trainable set (llamole_amd.distributed.allreduce_gradients)",
                      **{k: v for k, v in info.items() if k != "last_log"}},
           "tokens_per_s": ctx.world * B * S * args.steps / dt, "graph_side_ms": graph_ms, "graph_side_share": graph_ms / (1e3 * dt / args.steps),
           "loss": logd["loss"], "lm_loss": logd.get("lm_loss"), "retro_loss": logd.get("retro_loss"),
           "max_memory_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
           "llm_mfma": sft_llm_mfma(model, B, S, dt / args.steps),
           "roofline": roof}
    if not args.no_cpu_baseline and ctx.world == 1:
        log("cpu baseline ...")
        del model
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline_sft(args, usable_cores(), sd_pred, n_retro)
    return out


def _maybe_fail(rank: int, step: int):
    """Test hook (tests/test_bench_gpu.py): LLAMOLE_BENCH_FAIL_RANK=r makes rank r raise inside its first timed step, to check that the
    launcher ends the job with a non-zero exit instead of leaving the peers waiting in a collective."""
    r = os.environ.get("LLAMOLE_BENCH_FAIL_RANK")
    if r is not None and int(r) == rank and step == 0:
        raise RuntimeError(f"bench.py: injected failure on rank {rank} (LLAMOLE_BENCH_FAIL_RANK)")


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
    if args.batch is None:
        args.batch = 8 if (args.workload == "graphdit" or args.total_prompts) else args.targets if args.workload == "retro" else 1
    if args.workload == "retro":
        args.batch = args.targets

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
        out = run_retro(args, ctx) if args.workload == "retro" else run_sft(args, ctx)
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
    trace_on = os.environ.get("LLAMOLE_E2E_TRACE") == "1"      # host timeline of the e2e step (llamole_amd/_trace.py), folded on stderr
    if trace_on:
        from llamole_amd import _trace
        _trace.start()
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
        if insitu is not None and empty is not None:
            # PRICED WITH THIS RUN'S OWN MEASUREMENT (VERDICT r4 weak #3 / ADVICE r4): HIP events around each of the depth x T fc1 launches of one
            # launched trajectory on the idle GPU, minus what an event pair itself adds at that launch site (an empty pair recorded there in
            # a second trajectory).  The committed rocprofv3 trace average is a labelled cross-check below, never the priced figure.
            live = max(insitu[0] - empty[0], 1e-6)
            roof_dit = roofline_object(args, (live,) + tuple(dom_dit[1:]))
            roof_dit["timed"] = (f"this run, in situ: HIP events around each of the {insitu[1]} fc1 launches of one launched trajectory "
                                 f"({insitu[0] * 1e3:.2f} us) minus an empty event pair at the same launch sites ({empty[0] * 1e3:.2f} us)")
            roof_dit["kernel_ms_event_bracketed"] = insitu[0]
            roof_dit["event_pair_ms"] = empty[0]
        else:
            roof_dit = roofline_object(args, dom_dit)
            roof_dit["timed"] = "back to back over distinct weights (ll_gemm_bench): the trajectories of this run used another batch per engine call"
        roof_dit["kernel_ms_back_to_back"] = dom_dit[0]          # micro-benchmark: heads and tails of independent launches overlap (reads low)
        if marginal is not None:
            # what one fc1 launch adds to the dependent chain (trajectory with minus trajectory without the fc1 launches): kernel + launch boundary
            roof_dit["kernel_ms_marginal_in_chain"] = marginal[0]
            roof_dit["trajectory_ms_with_without_fc1"] = [marginal[1], marginal[2]]
        prof = graphdit_kernel_profile_avg(args, dit_batch)
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
        "roofline_graphdit": roof_dit,
        "rank_seconds": [round(t, 4) for t in rank_times],
        "host_threads_per_rank": torch.get_num_threads(),
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
