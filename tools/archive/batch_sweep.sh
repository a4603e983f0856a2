#!/bin/bash
# GPU box: e2e bench at several per-GPU batch sizes (robustness of the fused path for M = 2..4 and of the GEMM fallback beyond).
root=${GRAFT_REPO_ROOT:-$(pwd)}
for b in "$@"; do
  timeout 600 python3 $root/bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/bb.json
  python3 - "$b" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/bb.json").read())
    print("batch", sys.argv[1], "molecules/s", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 1))
except Exception as e:
    print("batch", sys.argv[1], "FAILED", repr(e))
PY
done
