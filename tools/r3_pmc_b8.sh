#!/bin/bash
# GPU box: the WRITE_SIZE pass at batch 8 alone (the FETCH / MfmaUtil CSVs of an earlier call are kept under profiles/raw on the host side);
# fallback: the two raw counters behind WRITE_SIZE in one pass.
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out
PMC_COUNTERS="WRITE_SIZE" SKIP_TRACE=1 bash "$root/tools/profile_step_pmc.sh" r3_graphdit_b8_step --batch 8 2>&1 | grep "attempt"
if [ ! -f "$out/r3_graphdit_b8_step_WRITE_SIZE.csv" ]; then
  export TMPDIR=/tmp; work=$(mktemp -d /tmp/wr.XXXXXX); cd "$work"
  for attempt in 1 2 3; do
    timeout 240 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d "$work/p$attempt" -o wr -- python3 "$root/bench.py" --workload graphdit --steps 1 --warmup 1 --no-graph --no-cpu-baseline --batch 8 > "$work/wr.log" 2>&1
    f=$(find "$work/p$attempt" -name '*counter_collection.csv' 2>/dev/null | head -1)
    if [ -n "$f" ] && grep -q '^{' "$work/wr.log"; then cp "$f" "$out/r3_graphdit_b8_step_TCC_EA0_WRREQ.csv"; echo "raw WRREQ pass ok"; break; fi
    echo "raw WRREQ attempt $attempt failed"
  done
fi
ls -la "$out"
