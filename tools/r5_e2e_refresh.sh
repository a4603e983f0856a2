#!/bin/bash
# GPU box: the headline line, its kernel trace and its device / host timelines at HEAD (copied from gpurun_out/ into profiles/ afterwards).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p "$out"
cd "$root"
bash tools/profile_bench.sh r5_e2e_b1 --steps 5 --warmup 2 > /dev/null 2>&1
python bench.py 2>/dev/null | grep '^{' > "$out/r5_bench_e2e.json"
python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
{ echo "# BENCH_VERBOSE=1 LLAMOLE_E2E_TRACE=2 python bench.py --steps 8 --warmup 2 --no-cpu-baseline   (pipelined: the production line)"
  BENCH_VERBOSE=1 LLAMOLE_E2E_TRACE=2 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep "device timeline\| ms  \|host timeline\|ms_per_step" | sed -e 's/^\[bench [0-9.]*s\] //' | cut -c1-200 | head -64
  echo "# the same with --no-pipeline (LLM decode, then the trajectory, per prompt)"
  BENCH_VERBOSE=1 LLAMOLE_E2E_TRACE=2 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipeline 2>&1 | grep "device timeline\| ms  \|ms_per_step" | sed -e 's/^\[bench [0-9.]*s\] //' | cut -c1-200 | head -24
} > "$out/r5_e2e_device_timeline.txt"
python bench.py --workload retro --steps 2 --warmup 1 2>/dev/null | grep '^{' > "$out/r5_bench_retro.json"
python - <<PY
import json
for f in ("r5_bench_e2e", "r5_bench_retro"):
    d = json.loads(open("$out/" + f + ".json").read().strip().splitlines()[-1])
    print(f, round(d["value"], 3), d["unit"], "ms/step", round(d["ms_per_step"], 2), "roof", round(d["roofline"]["frac"], 3), (d.get("cpu_baseline") or {}).get("value"))
PY
grep -i "gemv_fused_kernel\|gemv_stage" "$out"/r5_e2e_b1_kernel_stats.csv | head -5
