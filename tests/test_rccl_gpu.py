"""RCCL world-1 smoke: every collective the multi-GPU paths issue, driven through backend "nccl" (= RCCL on ROCm)
with ONE rank on the one GPU of the box -- process-group creation with ``device_id``, the device int8 all-gather of graph records, the
top-k all-gather, the float64 all-reduce of the timing, the bucketed gradient all-reduce, `bench.py` and `main.py eval` end to end --
so that an 8-GPU node is not RCCL's first contact with this code.  LLAMOLE_FORCE_DIST=1 disables the single-rank shortcuts
(llamole_amd/distributed.py:force_dist).  Multi-rank semantics are covered by the gloo tests (tests/test_distributed_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from llamole_amd import distributed as D
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and D.force_dist()
out = {}
# design phase: int8 records on the device
g = torch.Generator().manual_seed(0)
mols = []
for n in (5, 32, 1):
    e = torch.randint(0, 5, (n, n), generator=g)
    mols.append([torch.randint(0, 16, (n,), generator=g), (e + e.t()) %% 5])
got = D.all_gather_graphs(mols, 32, 3, device=dev)
out["graphs_equal"] = len(got) == 3 and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(got, mols))
# retro phase: candidate scores
idx = torch.randint(0, 1000, (4, 50), generator=g, dtype=torch.int32).to(dev)
prob = torch.rand(4, 50, generator=g).to(dev)
gi, gp = D.all_gather_topk(idx, prob)
out["topk_equal"] = bool(torch.equal(gi, idx) and torch.equal(gp, prob)) and gi.data_ptr() != idx.data_ptr()
# timing all-reduce (float64, MAX) and barrier
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
out["f64_max"] = float(t.item())
# gradient buckets (bf16 + f32 parameters, one parameter without a gradient)
ps = [torch.nn.Parameter(torch.randn(300, 40, device=dev, dtype=torch.bfloat16)), torch.nn.Parameter(torch.randn(77, device=dev)),
      torch.nn.Parameter(torch.randn(5, 5, device=dev))]
ps[0].grad = torch.ones_like(ps[0]); ps[1].grad = torch.full_like(ps[1], 2.0)
calls = D.allreduce_gradients(ps, bucket_bytes=16 << 10)
torch.cuda.synchronize()
out["grad_calls"] = calls
out["grads_kept"] = bool((ps[0].grad == 1).all() and (ps[1].grad == 2).all() and ps[2].grad is None)
dist.destroy_process_group()
print("RCCL1 " + json.dumps(out))
'''


def _env(port):
    return dict(os.environ, LLAMOLE_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_collectives_through_rccl_with_one_rank():
    p = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=_env(29721), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("RCCL1 ")][-1][6:])
    assert out["graphs_equal"] and out["topk_equal"] and out["f64_max"] == 1.25 and out["grads_kept"] and out["grad_calls"] >= 2, out


def test_bench_graphdit_through_rccl_with_one_rank():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "graphdit", "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--hidden", "256", "--depth", "2", "--heads", "4", "--T", "10", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=_env(29722), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["collectives"]["backend"] == "nccl" and line["collectives"]["forced_single_rank"]
    assert line["config"]["gathered_molecules"] == 8 and line["value"] > 0


def test_bench_sft_through_rccl_with_one_rank():
    """The SFT step's bucketed gradient all-reduce on device buckets under RCCL."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "sft", "--gpus", "1", "--steps", "1", "--warmup", "1", "--llm", "tiny",
           "--out-dim", "512", "--sft-batch", "2", "--sft-seq", "64", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=_env(29723), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0


def test_main_eval_through_rccl_with_one_rank(tmp_path):
    from llamole_amd import synth
    from llamole_amd.modeling_llamole import SPECIAL_TOKENS
    cfg = synth.write_eval_fixture(str(tmp_path), SPECIAL_TOKENS)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "eval_rank_worker.py"), cfg], env=_env(29724), capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    stats = json.loads([l for l in p.stdout.splitlines() if l.startswith("EVAL_STATS ")][-1][len("EVAL_STATS "):])
    assert stats["n_results"] == 5
