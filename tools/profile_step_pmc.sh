#!/bin/bash
# Run on the GPU box: whole-trajectory PMC passes over the GraphDiT-only workload (eager launches: counter collection does
# not see kernels inside a replayed hipGraph), ONE counter per pass, no trace domains.  Folded by tools/step_pmc_fold.py.
# usage: tools/profile_step_pmc.sh <tag> <bench args...>      e.g.  r1_graphdit_b8_step --batch 8
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in MfmaUtil FETCH_SIZE WRITE_SIZE; do
  rm -f /tmp/${tag}_${ctr}.csv
  for attempt in 1 2 3; do      # counter collection segfaults now and then on this image: retry the pass
    rm -rf /tmp/pmc_${tag}_$ctr
    timeout 90 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_${tag}_$ctr -o $tag -- python3 $root/bench.py --workload graphdit --steps 1 --warmup 1 --no-graph --no-cpu-baseline "$@" > $out/${tag}_${ctr}.log 2>&1
    f=$(find /tmp/pmc_${tag}_$ctr -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ] && grep -q '^{' $out/${tag}_${ctr}.log; then cp $f /tmp/${tag}_${ctr}.csv; break; fi
    echo "pass $ctr attempt $attempt failed"
  done
done
python3 $root/tools/step_pmc_fold.py $tag /tmp/${tag}_MfmaUtil.csv /tmp/${tag}_FETCH_SIZE.csv /tmp/${tag}_WRITE_SIZE.csv $out/${tag}_pmc.json "$@"
for ctr in MfmaUtil FETCH_SIZE WRITE_SIZE; do [ -f /tmp/${tag}_${ctr}.csv ] && rm -f $out/${tag}_${ctr}.log; done
