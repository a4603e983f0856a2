#!/bin/bash
# Run on the GPU box: one rocprofv3 --pmc pass per counter (no trace domains mixed in) over tools/gemm_one.py.
# usage: tools/profile_pmc.sh <tag> <counter> <M> [N K]
set -u
tag=$1; ctr=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
timeout 120 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$tag -o $tag -- python3 $root/tools/gemm_one.py "$@" > $out/${tag}_pmc.log 2>&1
tail -3 $out/${tag}_pmc.log | cut -c1-200
find /tmp/pmc_$tag -type f | head
f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && cp $f $out/${tag}_${ctr}.csv && head -3 $f | cut -c1-400
