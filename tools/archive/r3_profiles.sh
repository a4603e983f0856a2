#!/bin/bash
# GPU box: the round-3 bench lines and kernel statistics at HEAD (copied from gpurun_out/ into profiles/ afterwards).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p "$out"
cd "$root"
bash tools/profile_bench.sh r3_e2e_b1 --no-pipeline --steps 2 --warmup 1 > /dev/null 2>&1
bash tools/profile_bench.sh r3_graphdit_b1 --workload graphdit --batch 1 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/profile_bench.sh r3_graphdit_b8 --workload graphdit --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/profile_bench.sh r3_sft --workload sft --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/value_forward_profile.sh > "$out/r3_value_forward_probe.txt" 2>&1
python bench.py 2>/dev/null | grep '^{' > "$out/r3_bench_e2e.json"
python bench.py --no-pipeline --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r3_bench_e2e_nopipeline.json"
python bench.py --workload graphdit --steps 3 --warmup 1 2>/dev/null | grep '^{' > "$out/r3_bench_graphdit_b8.json"
python bench.py --workload graphdit --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r3_bench_graphdit_b1.json"
python bench.py --workload graphdit --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r3_bench_graphdit_b16.json"
python bench.py --workload retro 2>/dev/null | grep '^{' > "$out/r3_bench_retro.json"
python bench.py --workload sft 2>/dev/null | grep '^{' > "$out/r3_bench_sft.json"
python bench.py --llm llama-3.1-8b --total-prompts 64 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r3_bench_llama_total64_n1.json"
for f in r3_bench_e2e r3_bench_e2e_nopipeline r3_bench_graphdit_b8 r3_bench_graphdit_b1 r3_bench_graphdit_b16 r3_bench_retro r3_bench_sft r3_bench_llama_total64_n1; do python - <<PY
import json
try:
    d = json.loads(open("$out/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"], 3), d["unit"], "ms/step", round(d["ms_per_step"], 2), "dit_step_ms", round(d.get("denoise_step_ms") or 0, 4),
          "roof", round(d["roofline"]["frac"], 3), "roof_dit", round((d.get("roofline_graphdit") or {}).get("frac", 0) or 0, 3), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
