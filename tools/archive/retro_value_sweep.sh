#!/bin/bash
# GPU box: the retro workload (BASELINE configs[2]) against the two knobs of its A* value forwards -- prompts per LLM forward
# (LLAMOLE_VALUE_BATCH) and the shared-opening keys / values (LLAMOLE_VALUE_PREFIX=0 forwards every prompt whole).
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p "$out"; cd "$root"
for cfg in "256 8" "256 0" "512 8" "1024 8"; do
  set -- $cfg
  LLAMOLE_VALUE_BATCH=$1 LLAMOLE_VALUE_PREFIX=$2 python bench.py --workload retro --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/retro_sweep_$1_$2.json"
  python - "$out/retro_sweep_$1_$2.json" "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value_batch", sys.argv[2], "prefix_min", sys.argv[3], "->", round(d["value"], 4), d["unit"], round(d["ms_per_step"] / 1e3, 2), "s/step; value share",
      round(d["value_forward_share_of_step"], 3), "prompts/call", round(d["value_prompts_per_call"], 1), "opening", d["value_prompt_opening_tokens"])
PY
done
