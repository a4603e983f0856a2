#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace of bench.py for one workload, summarised to CSV.
# usage: tools/profile_bench.sh <tag> <bench args...>
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- python3 $root/bench.py "$@" --no-cpu-baseline > $out/${tag}_bench.log 2>&1
grep '^{' $out/${tag}_bench.log > $out/${tag}_bench.json
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
python3 $root/tools/rocpd_stats.py $db $out/${tag}_kernel_stats.csv $out/${tag}_kernel_gaps.csv > /dev/null
head -12 $out/${tag}_kernel_stats.csv
rm -f $out/${tag}_bench.log
