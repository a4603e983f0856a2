"""bench.py's rank launcher on a machine without GPUs: it must refuse (non-zero exit, message) before touching a device."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_refuses_without_enough_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2 but only" in r.stderr and not r.stdout.strip()
