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


def _run(extra, env_extra):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "graphdit", "--hidden", "128", "--depth", "2", "--heads", "4",
           "--T", "10", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


def test_bench_gpus_2_starts_two_ranks():
    r = _run(["--gpus", "2", "--batch", "3"], dict(LLAMOLE_BENCH_SHARED_GPU="1", LLAMOLE_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
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
