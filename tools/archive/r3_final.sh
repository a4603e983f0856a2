#!/bin/bash
# GPU box: the GPU test suite at HEAD, then the retro / sft / e2e bench lines of record (with cpu_baseline).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p "$out"
cd "$root"
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --workload retro 2>/dev/null | grep '^{' > "$out/r3_bench_retro.json"
python bench.py --workload retro --retro-constant-value --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r3_bench_retro_constant_value.json"
python bench.py --workload sft 2>/dev/null | grep '^{' > "$out/r3_bench_sft.json"
python bench.py 2>/dev/null | grep '^{' > "$out/r3_bench_e2e.json"
for f in r3_bench_e2e r3_bench_retro r3_bench_retro_constant_value r3_bench_sft; do python - <<PY
import json
try:
    d = json.loads(open("$out/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"], 3), d["unit"], "ms/step", round(d["ms_per_step"], 2), "roof", round(d["roofline"]["frac"], 3), (d.get("cpu_baseline") or {}).get("value"),
          d.get("expansions_per_s"), d.get("value_forward_share_of_step"), d.get("tokens_per_s"), d.get("graph_side_ms"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
