#!/bin/bash
# Run on the GPU box: whole-trajectory evidence for the GraphDiT-only workload at HEAD --
#   (1) rocprofv3 --kernel-trace --stats of `python3 bench.py --workload graphdit --no-graph ...` (kernel durations, launch gaps) and an
#       unprofiled run of the same command (the step time of record),
#   (2) PMC passes over the SAME command, ONE counter per pass and no trace domains (default: MfmaUtil, FETCH_SIZE, WRITE_SIZE; the raw
#       per-dispatch CSVs are kept as gpurun_out/<tag>_<counter>.csv so that a pass that crashed can be repeated alone:
#       PMC_COUNTERS="FETCH_SIZE" SKIP_TRACE=1 tools/profile_step_pmc.sh <tag> ...),
# folded by tools/step_pmc_fold.py into <tag>_pmc.json (per kernel and per reverse step: HBM-side bytes, GB/s, MFMA busy vs peak).
# `--no-graph`: the engine launches every kernel from its own loop -- counter collection does not see kernels inside a replayed hipGraph.
# TCC_EA0_RDREQ_sum is accepted in place of FETCH_SIZE (FETCH_SIZE = TCC_EA0_RDREQ x 64 B, MI355X_MICROARCH.md): the derived counter's
# pass segfaults rocprofv3 on some kernel mixes of this image.
# usage: tools/profile_step_pmc.sh <tag> <bench args...>      e.g.  r3_graphdit_b8_step --batch 8
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
work=$(mktemp -d /tmp/steppmc.XXXXXX)
cd "$work"
if [ -z "${SKIP_TRACE:-}" ]; then
  timeout 300 rocprofv3 --kernel-trace --stats -d "$work/kt" -o "$tag" -- python3 "$root/bench.py" --workload graphdit --steps 3 --warmup 1 --no-graph --no-cpu-baseline "$@" > "$work/kt.log" 2>&1
  db=$(find "$work/kt" -name '*.db' | head -1)
  [ -n "$db" ] && python3 "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_kernel_stats.csv" "$out/${tag}_kernel_gaps.csv" > /dev/null
  # the step time of record comes from an UNPROFILED run of the same command (tracing inflates it)
  python3 "$root/bench.py" --workload graphdit --steps 3 --warmup 1 --no-graph --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | tail -1 > "$out/${tag}_bench.json"
fi
for ctr in ${PMC_COUNTERS:-MfmaUtil FETCH_SIZE WRITE_SIZE}; do
  for attempt in 1 2 3; do      # counter collection segfaults now and then on this image: retry the pass in a fresh directory
    d="$work/pmc_${ctr}_$attempt"
    timeout 240 rocprofv3 --pmc $ctr --output-format csv -d "$d" -o "$tag" -- python3 "$root/bench.py" --workload graphdit --steps 1 --warmup 1 --no-graph --no-cpu-baseline "$@" > "$work/${ctr}.log" 2>&1
    f=$(find "$d" -name '*counter_collection.csv' 2>/dev/null | head -1)
    if [ -n "$f" ] && grep -q '^{' "$work/${ctr}.log"; then cp "$f" "$out/${tag}_${ctr}.csv"; break; fi
    echo "pass $ctr attempt $attempt failed"; tail -2 "$work/${ctr}.log" | cut -c1-200
  done
done
fetch="$out/${tag}_FETCH_SIZE.csv"
if [ ! -f "$fetch" ] && [ -z "${PMC_COUNTERS:-}" ]; then      # the derived counter's pass crashed every time: its raw counter instead
  for attempt in 1 2 3; do
    d="$work/pmc_rdreq_$attempt"
    timeout 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum --output-format csv -d "$d" -o "$tag" -- python3 "$root/bench.py" --workload graphdit --steps 1 --warmup 1 --no-graph --no-cpu-baseline "$@" > "$work/rdreq.log" 2>&1
    f=$(find "$d" -name '*counter_collection.csv' 2>/dev/null | head -1)
    if [ -n "$f" ] && grep -q '^{' "$work/rdreq.log"; then cp "$f" "$out/${tag}_TCC_EA0_RDREQ_sum.csv"; break; fi
    echo "pass TCC_EA0_RDREQ_sum attempt $attempt failed"; tail -2 "$work/rdreq.log" | cut -c1-200
  done
fi
[ -f "$fetch" ] || fetch="$out/${tag}_TCC_EA0_RDREQ_sum.csv"
python3 "$root/tools/step_pmc_fold.py" "$tag" "$out/${tag}_MfmaUtil.csv" "$fetch" "$out/${tag}_WRITE_SIZE.csv" "$out/${tag}_pmc.json" \
    "$out/${tag}_kernel_stats.csv" "$out/${tag}_bench.json" "$@"
