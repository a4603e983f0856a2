"""Eight ranks before an 8-GPU node sees this code (weak #10).

The driver's scaling run is the first time the paths below meet eight processes -- unless they are rehearsed here: eight ranks on the ONE GPU
of the test box (LLAMOLE_BENCH_SHARED_GPU=1: every rank uses device 0) talking over gloo (one device cannot host two RCCL ranks), toy
language model.  What eight ranks exercise that two do not: the launcher's port choice and the WorkQueue's MASTER_PORT + 17 store with eight
clients, shard arithmetic with remainders (64 prompts / 8, 16 targets / 8 with r, r + world, ... striding, 20 eval records over 8 claimants),
per-rank host thread shares (cores // 8), the fixed-size record gathers at world 8, and tearing eight ranks down when one of them dies."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHARED = dict(LLAMOLE_BENCH_SHARED_GPU="1", LLAMOLE_DIST_BACKEND="gloo")
TINY_DIT = ["--hidden", "128", "--depth", "2", "--heads", "4", "--T", "10"]
E2E_TINY = ["--llm", "tiny", "--new-tokens", "8", "--cutoff-len", "16", "--nodes", "16"]
RETRO_TINY = ["--llm", "tiny", "--iterations", "2", "--retro-tokens", "8", "--new-tokens", "8", "--cutoff-len", "16", "--nodes", "16",
              "--out-dim", "4096", "--topk", "10"]
SFT_TINY = ["--llm", "tiny", "--sft-batch", "2", "--sft-seq", "64", "--out-dim", "4096"]


def _bench(workload, extra, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LLAMOLE_BENCH_FAIL_RANK"):
        if k not in env_extra:
            env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload] + TINY_DIT + ["--steps", "1", "--warmup", "1", "--no-cpu-baseline"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def _line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _cores():
    sys.path.insert(0, ROOT)
    import bench
    return bench.usable_cores()


def test_e2e_64_prompts_over_8_ranks():
    """configs[3]'s shape on one GPU: 64 prompts per step, 8 ranks, batches of 8 -- one all-gather of 64 fixed-size graph records."""
    d = _line(_bench("e2e", ["--gpus", "8", "--total-prompts", "64", "--batch", "8"] + E2E_TINY, SHARED))
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and len(d["rank_seconds"]) == 8
    assert d["config"]["prompts_per_step"] == 64 and d["config"]["gathered_molecules"] == 64
    assert d["host_threads_per_rank"] == max(1, _cores() // 8)
    assert d["value"] > 0 and d["collectives"]["ranks"] == 8


def test_retro_16_targets_split_over_8_ranks_equals_solo():
    """--total-targets 16 over 8 ranks: the replicated lock-step A* with every round's expansions taken r, r + 8, ...; routes == one rank's."""
    one = _line(_bench("retro", ["--gpus", "1", "--total-targets", "16"] + RETRO_TINY, {}))
    eight = _line(_bench("retro", ["--gpus", "8", "--total-targets", "16"] + RETRO_TINY, SHARED))
    for d, n in ((one, 1), (eight, 8)):
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["config"]["prompts_per_step"] == 16 and d["config"]["gathered_routes"] == 16
        assert d["expansions"] == 16 * 2                                       # counted once, not once per rank
    assert one["routes_found"] == eight["routes_found"] and one["route_lengths"] == eight["route_lengths"]
    assert "split over 8 GPU(s)" in eight["config"]["workload"] and eight["host_threads_per_rank"] == max(1, _cores() // 8)


def test_sft_data_parallel_8_ranks():
    d = _line(_bench("sft", ["--gpus", "8"] + SFT_TINY, SHARED))
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp8"
    assert d["loss"] == d["loss"] and d["retro_loss"] > 0 and d["value"] > 0


def test_main_eval_8_ranks_on_the_work_queue(tmp_path):
    """`main.py eval` with eight claimants on the TCPStore counter (MASTER_PORT + 17): 20 records in batches of 2, every record exactly once,
    gathered in order on rank 0."""
    from llamole_amd import synth
    from llamole_amd.modeling_llamole import SPECIAL_TOKENS
    cfg = synth.write_eval_fixture(str(tmp_path), SPECIAL_TOKENS)
    synth.write_molqa_dataset(os.path.join(str(tmp_path), "data"), n=20)
    port = None
    for _ in range(50):          # a rendezvous port whose WorkQueue neighbour (MASTER_PORT + 17, distributed.WorkQueue) is free as well
        a, b = socket.socket(), socket.socket()
        try:
            a.bind(("127.0.0.1", 0))
            b.bind(("127.0.0.1", a.getsockname()[1] + 17))
            port = a.getsockname()[1]
        except OSError:
            pass
        finally:
            a.close()
            b.close()
        if port is not None:
            break
    assert port is not None
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **SHARED)
        env.pop("LLAMOLE_QUEUE_PORT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "eval_rank_worker.py"), cfg], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    line = [l for l in outs[0][0].splitlines() if l.startswith("EVAL_STATS ")][-1]
    summary = json.loads(line[len("EVAL_STATS "):])
    assert summary["n_results"] == 20
    saved = json.load(open(os.path.join(str(tmp_path), "out", "molqa_results.json")))
    assert [r["qa_idx"] for r in saved] == list(range(20))


def test_rank_failure_at_world_8_ends_the_job():
    t0 = time.time()
    r = _bench("e2e", ["--gpus", "8", "--batch", "2", "--steps", "2"] + E2E_TINY, dict(SHARED, LLAMOLE_BENCH_FAIL_RANK="5"), timeout=600)
    assert r.returncode != 0 and "injected failure" in r.stderr and "rank 5 exited" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 560
