#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace of the GIN predictor / encoder legs of tools/gin_bench.py, summarised.
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for leg in predictor encoder; do
  rm -rf /tmp/prof_gin_$leg
  rocprofv3 --kernel-trace --stats -d /tmp/prof_gin_$leg -o gin -- python3 $root/tools/gin_bench.py --no-cpu --only $leg > /dev/null 2>&1
  db=$(find /tmp/prof_gin_$leg -name "*.db" | head -1)
  python3 $root/tools/rocpd_stats.py $db $out/r3_gin_${leg}_kernel_stats.csv $out/r3_gin_${leg}_kernel_gaps.csv > /dev/null
  (cd $root/tools && python3 rocpd_timeline.py $db graph_csr_kernel - $out/r3_gin_${leg}_timeline.csv) | tail -3
done
cd $root && python tools/gin_bench.py > $out/r3_gin_bench.json 2>/dev/null; cut -c1-600 $out/r3_gin_bench.json
