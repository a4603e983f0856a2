#!/usr/bin/env python3
"""Benchmark of the Llamole interleaved-generation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload graphdit|e2e]

One "step" = one pass of the hot path over one batch of synthetic prompts+conditions:
  * graphdit : B property/text conditions -> full T-step GraphDiT reverse diffusion -> B integer
               molecule graphs (BASELINE.json configs[0] shape at the reference-default denoiser size);
  * e2e      : BASELINE.json configs[1]: Qwen2-7B (random-init, HF on PyTorch-ROCm) decodes to the
               design trigger, query-token re-forward, connector, then the GraphDiT trajectory.
Prints ONE JSON line (rank 0).  `value` = molecules/s over all ranks (weak scaling: every rank runs
its own batch of independent prompts; no data-path collective, one small all-gather of the graphs).
"""
import argparse
import json
import os
import sys
import time

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
        H, Hm = args.hidden, int(args.hidden * 4)
        M, N, K = 2 * args.batch * args.nodes, Hm, H
        kern = ("gemm_m64_kernel<8,8,bf16,packed>" if M <= 64 else "gemm_bf16_pipeu_kernel<64,64,4,4,4>" if M < 1024 else
                "gemm_bf16_pipe_kernel<128,128,4,4,3>" if M < 2048 else "gemm_bf16_pipe_kernel<256,128,4,4,3>")
        name = f"{kern}, GraphDiT block-MLP fc1 [{M}x{K}]x[{N}x{K}]^T bf16"
        key = f"fc1_m{M}"
    nw = max(2, int(600e6 // (N * K * 2)))
    ms = C.c_float()
    _lib.check(lib.ll_gemm_bench(M, N, K, -1, 1, 0, 4 * nw, nw, C.byref(ms)), "ll_gemm_bench")
    nbytes = N * K * 2 + M * K * 2 + M * N * 2
    flops = 2.0 * M * N * K
    return ms.value, nbytes, flops, name, key


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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="e2e", choices=["graphdit", "e2e"])
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
    ap.add_argument("--llm", default="qwen2-7b")
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
    global GRAPH_MODE
    GRAPH_MODE = False if args.no_graph else (True if args.graph else None)      # None: the library's choice
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this parent starts the N ranks itself, BEFORE it makes any GPU call
        # (a process that has initialised the GPU must not exec/replace itself on this pool; device_count() does not initialise)
        raise SystemExit(spawn_ranks(args.gpus))
    if args.batch is None:
        args.batch = 8 if (args.workload == "graphdit" or args.total_prompts) else 1

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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LLAMOLE_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; "gloo" only for single-GPU dry runs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

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
    t0 = time.perf_counter()
    dit_ms = []
    piped = bool(getattr(step_fn, "pipeline", False))
    if piped:
        del step_fn.dit_ms[:]
    done = []                            # molecules this rank produced in the timed region, in prompt order
    for i in range(args.steps * nb):
        mols = step_fn(args.warmup * nb + i)
        if not piped:
            dit_ms.append(m.last_run_ms()[0])
        done.extend(mols or [])
    if piped:
        done.extend(step_fn.finish() or [])    # the last batch's trajectory completes inside the timed region
        dit_ms = list(step_fn.dit_ms)[-args.steps * nb:]
    barrier()
    dt = time.perf_counter() - t0
    assert len(done) == args.steps * nb * B, (len(done), args.steps, nb, B)
    gathered_n = len(done[-nb * B:])
    if dist is not None:
        tmax = torch.tensor([dt], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
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
    kms, kbytes, kflops, kname, kkey = dom
    log("dominant kernel timed", kms)
    hbm_t, mfma_t = kbytes / (HBM_PEAK_GBS * 1e9), kflops / (MFMA_BF16_PEAK_TF * 1e12)
    if hbm_t >= mfma_t or args.dtype != "bf16":
        roof = {"bound": "hbm", "achieved": kbytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    else:
        roof = {"bound": "mfma", "achieved": kflops / (kms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"] = None
    try:   # HBM bytes per launch from the PMC passes committed under profiles/ (same kernel, same shape)
        pmc_file = "r2_pmc_traffic.json"       # r1 entries (LLM kernels, frozen since) + the GraphDiT fc1 GEMM re-measured at HEAD
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
        key = kkey
        if key in pmc and args.dtype == "bf16" and (not key.startswith("fc1") or args.hidden == 1024):
            roof["traffic"] = pmc[key]["hbm_bytes_per_launch"]
            roof["traffic_source"] = f"profiles/{pmc_file} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, FETCH doubled per MI355X_MICROARCH.md)"
    except Exception:
        pass
    roof["algorithmic_bytes"] = kbytes
    roof["kernel"] = kname
    roof["kernel_ms"] = kms
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
                   "dit_launch": ("launches" if args.no_graph else "graph" if args.graph else "auto (launches alone, graph replay when overlapped with the LLM)"), **e2e_info},
        "denoise_steps_per_s": world * 1e3 / step_ms,
        "denoise_step_ms": step_ms,
        "denoise_step_ms_overlapped_with_llm": step_ms_overlapped,
        "step_roofline": {"hbm_bytes": sbytes, "flops": sflops,
                          "hbm_frac": sbytes / (step_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                          "mfma_frac": sflops / (step_ms * 1e-3) / (MFMA_BF16_PEAK_TF * 1e12)},
        "roofline": roof,
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
