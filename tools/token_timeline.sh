#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace of the headline bench (optionally --no-pipeline), then the device timeline of single decode tokens
# (anchor: decode_prologue_kernel) at a few places of the run.  usage: tools/token_timeline.sh <tag> [bench args]
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/timeline
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace -d /tmp/tl_$tag -o $tag -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $out/${tag}_bench.log 2>&1
grep '^{' $out/${tag}_bench.log | cut -c1-300
db=$(find /tmp/tl_$tag -name '*.db' | head -1)
for k in 500; do
    python3 $root/tools/rocpd_timeline.py $db decode_prologue $k $out/${tag}_token_$k.csv | tail -1
done
python3 $root/tools/token_periods.py $db > $out/${tag}_periods.txt; cat $out/${tag}_periods.txt
