"""Workloads, timing helpers and roofline objects behind ``bench.py`` (split out in round 5: bench.py keeps the driver contract --
argument parsing, rank launch, the timed region of the headline workloads, the JSON line -- and the CPU-baseline legs, which are the
only code of this repository besides tests/ and __graft_entry__.smoke() allowed to touch ``oracle/``; nothing in this module does).

  * model / weight construction for the synthetic configs (``build_model``, ``fast_dit_weights``), host-core accounting (``usable_cores``);
  * algorithmic bytes / flops of a GraphDiT reverse step (``dit_step_bytes`` / ``dit_step_flops``);
  * kernel timing: the dominant hand-written kernel of each workload with HIP events on the stream it runs on (``time_dominant_kernel``,
    ``time_template_head``), the GraphDiT fc1 GEMM in situ (event bracket, empty-pair calibration, marginal cost), ``roofline_object``;
  * the retro (BASELINE configs[2]) and sft (configs[4]) workloads as ``run_retro`` / ``run_sft`` (the CPU baseline is handed in by bench.py).
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch

from . import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLM_LABEL = {"qwen2-7b": "Qwen2-7B", "llama-3.1-8b": "Llama-3.1-8B", "mistral-7b": "Mistral-7B"}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0
_T0 = time.perf_counter()


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
        if args.llm_fuse and args.llm_layer_fuse and args.llm_decode != "hf" and 16 < M <= 64 and K % 32 == 0 and N % 16 == 0:
            # 17..64 sequences: the same projection on the packed-weight stream (ll_linear_rows64_bf16; its RMSNorm input comes from
            # the preceding reduce launch, so the kernel itself is GEMM + SiLU*mul)
            rows = 2 * N
            nw = max(2, int(600e6 // (rows * K * 2)))
            _lib.check(lib.ll_rows64_bench(M, N, K, 2, 0, 8 * nw, nw, C.byref(ms)), "ll_rows64_bench")
            name = (f"rows64_kernel<silu_mul,{2 if M <= 32 else 4}>, LLM gated-MLP gate|up projection [{M}x{K}]x[{rows}x{K}]^T bf16 on weights in "
                    f"MFMA operand order + SiLU*mul epilogue (batched decode step)")
            nbytes = rows * K * 2 + M * K * 2 + M * N * 2
            return ms.value, nbytes, 2.0 * M * rows * K, name, f"llm_rows64_m{M}_n{rows}_k{K}"
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


def token_roofline(args, orch, prompt, mask, gen_kw):
    """The WHOLE decode token against the HBM roofline (the line's `roofline` is its best kernel, not the token):
    bytes = every weight matrix a decode step streams (q|k|v, o_proj, gate|up, down_proj of every layer + lm_head; the KV cache and the
    activations are not counted), time = device time per token of one more `generate` on the idle GPU after the timed region, from HIP
    events recorded on the LLM's stream at the decode loop's marks (first token sampled -> decode loop done).  Next to it every Linear of
    the step timed back to back over distinct weights (HIP events; the same micro-benchmark entry points as `roofline`), at this run's
    row count, so that the share the weight stream does NOT explain -- attention, sampler, the launch chain -- is visible."""
    import ctypes as C
    from llamole_amd import _lib, _trace
    from llamole_amd.e2e import LLM_CONFIGS
    if args.llm_linear != "hip" or args.llm_decode == "hf" or getattr(orch, "decoder", None) is None:
        return None
    lib = _lib.load()
    spec = LLM_CONFIGS[args.llm]
    H, I, V, L = spec["hidden_size"], spec["intermediate_size"], spec["vocab_size"], spec["num_hidden_layers"]
    D = H // spec["num_attention_heads"]
    nq, nkv = spec["num_attention_heads"] * D, spec["num_key_value_heads"] * D
    M = int(prompt.shape[0])
    shapes = {"q|k|v": (nq + 2 * nkv, H, 0), "o_proj": (H, nq, 1), "gate|up": (I, H, 2), "down_proj": (H, I, 1), "lm_head": (V, H, 0)}
    ms = C.c_float()
    per = {}
    fused = bool(args.llm_fuse and args.llm_layer_fuse)
    for name, (N, K, epi) in shapes.items():
        rows = 2 * N if epi == 2 else N
        nbytes = rows * K * 2
        nw = max(2, int(600e6 // nbytes))
        norm = 1 if name in ("q|k|v", "gate|up") else 0           # RMSNorm prologue of the five-launch layer
        try:
            if fused and 16 < M <= 64 and K % 32 == 0 and (epi != 2 or N % 16 == 0):
                # seven-launch layer: q|k|v / gate|up finish a split RMSNorm (row scale), o_proj / down_proj produce the next one (+ slab sum)
                flag = 0 if name == "lm_head" else (2 if norm else 1)
                _lib.check(lib.ll_rows64_bench(M, N, K, epi, flag, 4 * nw, nw, C.byref(ms)), "ll_rows64_bench")
                kern = "rows64_kernel" + (" + rows64_reduce_kernel" if flag == 1 else "")
            elif fused and 3 <= M <= 16 and K % 32 == 0:
                _lib.check(lib.ll_rows16_bench(M, N, K, epi, norm if name != "lm_head" else 0, 4 * nw, nw, C.byref(ms)), "ll_rows16_bench")
                kern = "rows16_kernel"
            elif fused and M <= 4 and name != "lm_head":
                _lib.check(lib.ll_gemv_fused_bench(M, N, K, epi, norm, 1, 4 * nw, nw, C.byref(ms)), "ll_gemv_fused_bench")
                kern = "gemv_fused_kernel / gemv_stage_kernel"
            else:
                _lib.check(lib.ll_gemm_bench(M, rows, K, -1, 1, 0, 4 * nw, nw, C.byref(ms)), "ll_gemm_bench")
                kern = "ll_linear dispatch"
        except RuntimeError as e:       # a shape the micro-benchmark refuses: the token figure below does not depend on it
            per[name] = {"error": str(e)}
            continue
        per[name] = {"kernel": kern, "bytes": nbytes, "ms": ms.value, "gbps": nbytes / (ms.value * 1e-3) / 1e9,
                     "frac": nbytes / (ms.value * 1e-3) / (HBM_PEAK_GBS * 1e9), "launches_per_token": 1 if name == "lm_head" else L}
    bytes_token = sum(v["bytes"] * v["launches_per_token"] for v in per.values() if "bytes" in v)
    lin_ms = sum(v["ms"] * v["launches_per_token"] for v in per.values() if "ms" in v)
    # one more generation on the idle GPU with a HIP event at every mark of the decode loop
    torch.cuda.synchronize()
    _trace.start(device=True)
    try:
        torch.manual_seed(4242)
        orch._llm_generate(inputs=prompt, attention_mask=mask, **gen_kw)
        torch.cuda.synchronize()
    finally:
        ev = _trace.stop()
    dev = ev[1] if isinstance(ev, tuple) else []
    first = next((e for n, e in dev if n == "generate: first token sampled"), None)
    done = next((e for n, e in dev if n == "generate: decode loop done"), None)
    n_new = int(getattr(orch.decoder, "_last", {}).get("n_new", 0) or 0)
    if first is None or done is None or n_new < 2:
        return {"per_linear": per, "bytes_per_token": bytes_token, "note": "the decode loop's marks were not recorded"}
    token_ms = first.elapsed_time(done) / (n_new - 1)
    return {"bound": "hbm", "bytes_per_token": bytes_token, "token_ms": token_ms, "tokens_timed": n_new - 1, "rows": M,
            "achieved": bytes_token / (token_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_token / (token_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
            "timed": "HIP events on the LLM stream around the decode loop of one generation on the idle GPU after the timed region",
            "linears_ms_per_token_back_to_back": lin_ms, "not_weight_stream_ms_per_token": token_ms - lin_ms,
            "per_linear": per}


def time_graphdit_kernel(args, batch: int):
    """The dominant kernel of the GraphDiT step at `batch` graphs: the block-MLP fc1 GEMM at M = 2 * batch * N token rows, on whatever
    kernel the production dispatch picks for that M (32-33 % of the step's GPU time in profiles/r*_graphdit_b{1,8}_kernel_stats.csv),
    timed by HIP events over distinct weight matrices (ll_gemm_bench).  Returns (avg_ms, algorithmic bytes, flops, name, pmc key)."""
    import ctypes as C
    from llamole_amd import _lib
    lib = _lib.load()
    H, Hm = args.hidden, int(args.hidden * 4)
    M, N, K = 2 * batch * args.nodes, Hm, H
    panel64 = K in (256, 512, 768, 1024)          # K chunks the all-in-flight panel kernels exist for (gemm.hip: gemm_dispatch)
    panel = K in (256, 512, 1024)
    kern = (f"gemm_m64_kernel<{K // 128},{8 if (K // 128) % 2 == 0 else 4},bf16,packed>" if M <= 64 and panel64 else "gemm_m128_kernel" if 64 < M <= 224 and panel else
            "gemm_bf16_pipe_kernel<32,32,2,2,8>" if M <= 32 else
            "gemm_bf16_pipeu_kernel<64,64,4,4,4>" if 64 < M < 1024 else "gemm_bf16_pipe_kernel<64,*> (LDS-DMA ring)" if M <= 64 else
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
    # newest first (r6 holds only what round 6 measured; r5 carried round-3 entries over under its own name, so an entry's own `method`
    # field says which round it is from -- printed in traffic_source)
    for pmc_file in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
        except Exception:
            continue
        if kkey in pmc and args.dtype == "bf16" and (not kkey.startswith("fc1") or args.hidden == 1024):
            roof["traffic"] = pmc[kkey]["hbm_bytes_per_launch"]
            meth = str(pmc[kkey].get("method", ""))
            rnd = next((r for r in ("r6", "r5", "r4", "r3", "r2", "r1") if f"({r})" in meth), None)
            roof["traffic_source"] = (f"profiles/{pmc_file}" + (f", entry measured in round {rnd[1:]}" if rnd else ", entry carried over from an earlier round")
                                      + " (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes, FETCH doubled per MI355X_MICROARCH.md)")
            break
    roof["algorithmic_bytes"] = kbytes
    roof["algorithmic_flops"] = kflops
    roof["kernel"] = kname
    roof["kernel_ms"] = kms
    return roof


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
        pmc_name = "r5_pmc_traffic.json" if os.path.exists(os.path.join(ROOT, "profiles", "r5_pmc_traffic.json")) else "r3_pmc_traffic.json"
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
        key = f"gin_head_rows16_m{M}_n{N}_k{K}"
        if key in pmc:
            roof["traffic"] = pmc[key]["hbm_bytes_per_launch"]
            roof["traffic_source"] = f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, FETCH doubled per MI355X_MICROARCH.md)"
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


def run_retro(args, ctx, cpu_baseline_fn=None):
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
        if cpu_baseline_fn is not None:
            out["cpu_baseline"] = cpu_baseline_fn(args, llm, sd_pred, usable_cores(), n_val / max(1, n_exp))
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


def run_sft(args, ctx, cpu_baseline_fn=None):
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
        if cpu_baseline_fn is not None:
            out["cpu_baseline"] = cpu_baseline_fn(args, usable_cores(), sd_pred, n_retro)
    return out


def _maybe_fail(rank: int, step: int):
    """Test hook (tests/test_bench_gpu.py): LLAMOLE_BENCH_FAIL_RANK=r makes rank r raise inside its first timed step, to check that the
    launcher ends the job with a non-zero exit instead of leaving the peers waiting in a collective."""
    r = os.environ.get("LLAMOLE_BENCH_FAIL_RANK")
    if r is not None and int(r) == rank and step == 0:
        raise RuntimeError(f"bench.py: injected failure on rank {rank} (LLAMOLE_BENCH_FAIL_RANK)")
