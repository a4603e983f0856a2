"""bench.py's own launcher: `python bench.py --gpus 2` must start two ranks by itself, connect them through
torch.distributed and report what the collective saw.  On the one-GPU test box both ranks share device 0 and talk over gloo
(LLAMOLE_BENCH_SHARED_GPU=1 / LLAMOLE_DIST_BACKEND=gloo: dry-run switches of bench.py, RCCL needs one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SHARED = dict(LLAMOLE_BENCH_SHARED_GPU="1", LLAMOLE_DIST_BACKEND="gloo")
TINY_DIT = ["--hidden", "128", "--depth", "2", "--heads", "4", "--T", "10"]


def _run(extra, env_extra, workload="graphdit", timeout=600):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LLAMOLE_BENCH_FAIL_RANK"):
        if k not in env_extra:
            env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload] + TINY_DIT + ["--steps", "1", "--warmup", "1",
                                                                                                 "--no-cpu-baseline"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def _line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks():
    r = _run(["--gpus", "2", "--batch", "3"], dict(LLAMOLE_BENCH_SHARED_GPU="1", LLAMOLE_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and len(d["rank_seconds"]) == 2 and max(d["rank_seconds"]) * 1e3 / d["steps"] == pytest.approx(d["ms_per_step"], rel=5e-2)
    assert d["config"]["gathered_molecules"] == 2 * 3 and d["config"]["prompts_per_step"] == 6
    assert d["value"] > 0 and "roofline" in d


def test_bench_total_prompts_is_strong_scaling():
    r = _run(["--gpus", "2", "--batch", "2", "--total-prompts", "8"], dict(LLAMOLE_BENCH_SHARED_GPU="1", LLAMOLE_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["prompts_per_step"] == 8 and d["config"]["gathered_molecules"] == 8


def test_bench_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(n)], {})
    assert r.returncode != 0 and "visible" in r.stderr


# ---- the e2e path under world > 1: pipelined trajectories, finish() draining and the gather, before an 8-GPU node runs it
E2E_TINY = ["--llm", "tiny", "--new-tokens", "8", "--cutoff-len", "16", "--nodes", "16"]


def test_bench_e2e_two_ranks_weak():
    d = _line(_run(["--gpus", "2", "--batch", "2", "--steps", "3"] + E2E_TINY, SHARED, workload="e2e"))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["prompts_per_step"] == 4 and d["config"]["gathered_molecules"] == 4
    assert "side HIP stream" in d["config"]["pipeline"] and d["value"] > 0
    assert d["denoise_step_ms_overlapped_with_llm"] is not None and "roofline" in d and "roofline_graphdit" in d


def test_bench_e2e_two_ranks_total_prompts():
    d = _line(_run(["--gpus", "2", "--batch", "2", "--total-prompts", "8", "--steps", "2"] + E2E_TINY, SHARED, workload="e2e"))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["prompts_per_step"] == 8 and d["config"]["gathered_molecules"] == 8


def test_bench_total_prompts_picks_the_whole_share_as_one_batch():
    """without --batch a rank decodes its share of --total-prompts together, up to the 64 sequences the fused decode
    path serves (ll_linear_rows64_bf16 + the seven-launch layer), and the line says which batch ran and what an N = 1 baseline means."""
    d = _line(_run(["--total-prompts", "48", "--steps", "2"] + E2E_TINY, {}, workload="e2e"))
    c = d["config"]
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and c["prompts_per_step"] == 48 and c["gathered_molecules"] == 48
    assert c["per_rank_batch"] == 48 and c["per_rank_batches_per_step"] == 1 and "min(share" in c["per_rank_batch_source"]
    assert "BEST single-GPU configuration" in c["scaling_baseline"]
    assert "rows64_kernel" in d["roofline"]["kernel"] and d["roofline"]["bound"] == "hbm"
    rt = d["roofline_token"]        # the whole decode token against the weight stream, with the per-Linear fractions of this run
    assert 0 < rt["frac"] < 1 and rt["rows"] == 48 and rt["tokens_timed"] == 7 and set(rt["per_linear"]) == {"q|k|v", "o_proj", "gate|up", "down_proj", "lm_head"}
    assert all("rows64_kernel" in v["kernel"] for v in rt["per_linear"].values()) and d["library"] == "libllamole_hip_tuning.so"
    assert c["llm_fused_elementwise"]["decoder_layers_5_launches"] == 2
    # 128 prompts on one rank: two batches of 64; --batch still wins when given
    d = _line(_run(["--total-prompts", "128", "--steps", "1"] + E2E_TINY, {}, workload="e2e"))
    assert d["config"]["per_rank_batch"] == 64 and d["config"]["per_rank_batches_per_step"] == 2 and d["config"]["gathered_molecules"] == 128
    d = _line(_run(["--total-prompts", "16", "--batch", "4", "--steps", "1"] + E2E_TINY, {}, workload="e2e"))
    assert d["config"]["per_rank_batch"] == 4 and d["config"]["per_rank_batches_per_step"] == 4 and d["config"]["per_rank_batch_source"] == "--batch"
    # two ranks (sharing this box's GPU over gloo): 24 sequences each on the seven-launch layers, one gather of 48 molecules
    d = _line(_run(["--gpus", "2", "--total-prompts", "48", "--steps", "1"] + E2E_TINY, SHARED, workload="e2e"))
    assert d["n_gpus"] == 2 and d["config"]["per_rank_batch"] == 24 and d["config"]["gathered_molecules"] == 48 and "rows64_kernel" in d["roofline"]["kernel"]


def test_bench_rank_failure_ends_the_job():
    """One rank raising inside the timed region: the launcher must end every rank and exit non-zero -- not hang in the barrier."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--batch", "2", "--steps", "2"] + E2E_TINY, dict(SHARED, LLAMOLE_BENCH_FAIL_RANK="1"), workload="e2e", timeout=300)
    assert r.returncode != 0 and "injected failure" in r.stderr and "rank 1 exited" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 280


# ---- configs[2] / configs[4] workloads at toy sizes: one rank and two
RETRO_TINY = ["--llm", "tiny", "--targets", "3", "--iterations", "2", "--retro-tokens", "8", "--new-tokens", "8", "--cutoff-len", "16",
              "--nodes", "16", "--out-dim", "4096", "--topk", "10"]
SFT_TINY = ["--llm", "tiny", "--sft-batch", "2", "--sft-seq", "64", "--out-dim", "4096"]


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_retro_workload(gpus):
    d = _line(_run(["--gpus", str(gpus)] + RETRO_TINY, SHARED if gpus > 1 else {}, workload="retro"))
    assert d["n_gpus"] == gpus and d["unit"] == "molecules/s" and d["value"] > 0
    assert d["config"]["prompts_per_step"] == 3 * gpus and d["config"]["gathered_routes"] == 3 * gpus
    assert d["expansions"] == 3 * 2 * gpus and d["expansions_per_s"] > 0           # every search runs its 2 expansions (one closes at the second)
    assert d["routes_found"] == gpus and d["route_lengths"] == [2] * gpus and d["searches_without_route"] == 2 * gpus      # scripted chemistry: S2 / U / S3
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0 and "A* retrosynthesis" in d["config"]["workload"]
    # the value forwards: all searches of a round in one call, the prompts' shared opening forwarded once, their rate against the MFMA peak
    assert d["value_prompts_per_call"] > 3 and d["value_prompt_opening_tokens"] >= 8
    assert d["value_forward_mfma"]["bound"] == "mfma" and 0 < d["value_forward_mfma"]["frac"] < 1 and d["value_forward_mfma"]["tokens"] > 0


def test_bench_retro_strong_scaling_splits_every_round():
    """--total-targets: ONE lock-step A* for the whole job, replicated on every rank, each round's expansions and value prompts split over
    the ranks (one all-gather of top-k records + analysis tokens, one of the costs): the routes of two ranks equal those of one."""
    one = _line(_run(["--gpus", "1", "--total-targets", "4"] + RETRO_TINY, {}, workload="retro"))
    two = _line(_run(["--gpus", "2", "--total-targets", "4"] + RETRO_TINY, SHARED, workload="retro"))
    for d, n in ((one, 1), (two, 2)):
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["config"]["prompts_per_step"] == 4 and d["config"]["gathered_routes"] == 4
        assert d["expansions"] == 4 * 2                                            # counted once, not once per rank
    assert one["routes_found"] == two["routes_found"] == 1 and one["route_lengths"] == two["route_lengths"] == [2]
    assert "split over 2 GPU(s)" in two["config"]["workload"]


def test_bench_retro_constant_value_shortcut():
    d = _line(_run(["--retro-constant-value"] + RETRO_TINY, {}, workload="retro"))
    assert d["expansions"] == 3 * 2 and "shortcut" in d["config"]["value_estimates"] and d["value_forward_share_of_step"] < 0.2
    assert d["value_forward_mfma"] is None                                          # no forward ran


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_sft_workload(gpus):
    d = _line(_run(["--gpus", str(gpus)] + SFT_TINY, SHARED if gpus > 1 else {}, workload="sft"))
    assert d["n_gpus"] == gpus and d["unit"] == "samples/s" and d["value"] > 0
    assert d["config"]["global_batch"] == 2 * gpus and d["config"]["parallelism"] == f"dp{gpus}"
    assert d["loss"] == d["loss"] and d["retro_loss"] > 0 and d["graph_side_ms"] > 0 and d["roofline"]["frac"] > 0
    assert d["llm_mfma"]["bound"] == "mfma" and 0 < d["llm_mfma"]["frac"] < 1


def test_bench_under_torch_distributed_run():
    """The driver's own launch line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N --steps K --warmup W` -- ranks from the launcher's environment, ONE JSON line from rank 0."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, **SHARED)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LLAMOLE_BENCH_FAIL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--batch", "2"] + TINY_DIT + E2E_TINY
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["config"]["gathered_molecules"] == 4
