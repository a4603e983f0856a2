#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd $root
bash tools/profile_bench.sh r2_graphdit_b1 --workload graphdit --batch 1 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/profile_bench.sh r2_graphdit_b8 --workload graphdit --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/profile_bench.sh r2_e2e_b1 --no-pipeline --steps 2 --warmup 1 > /dev/null 2>&1
python bench.py 2>/dev/null | grep '^{' > $out/r2_bench_e2e.json
python bench.py --workload graphdit --steps 3 --warmup 1 2>/dev/null | grep '^{' > $out/r2_bench_graphdit_b8.json
python bench.py --workload graphdit --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/r2_bench_graphdit_b1.json
python bench.py --no-pipeline --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/r2_bench_e2e_nopipeline.json
python bench.py --llm llama-3.1-8b --total-prompts 64 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/r2_bench_llama_total64_n1.json
python tools/gin_bench.py > $out/r2_gin_bench.json 2>/dev/null
python tools/retro_bench.py > $out/r2_retro_bench.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_gin -o gin -- python3 $root/tools/gin_bench.py --no-cpu > /dev/null 2>&1
python3 $root/tools/rocpd_stats.py $(find /tmp/prof_gin -name "*.db" | head -1) $out/r2_gin_kernel_stats.csv $out/r2_gin_kernel_gaps.csv > /dev/null
cd $root; for f in r2_bench_e2e r2_bench_graphdit_b8 r2_bench_graphdit_b1 r2_bench_e2e_nopipeline r2_bench_llama_total64_n1; do python - <<PY
import json
try:
    d=json.loads(open("$out/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"],3), d["unit"], "ms/step", round(d["ms_per_step"],2), "dit_step_ms", round(d.get("denoise_step_ms") or 0,4), "roof", round(d["roofline"]["frac"],3), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print("$f", "ERR", e)
PY
done
head -3 $out/r2_gin_bench.json | cut -c1-400; cut -c1-500 $out/r2_retro_bench.json
