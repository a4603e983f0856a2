#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p "$out"
export TMPDIR=/tmp; work=$(mktemp -d /tmp/vf.XXXXXX); cd "$work"
python3 "$root/tools/value_forward_probe.py" qwen2-7b 1024
rocprofv3 --kernel-trace --stats -d "$work/kt" -o vf -- python3 "$root/tools/value_forward_probe.py" qwen2-7b 1024 default > /dev/null 2>&1
db=$(find "$work/kt" -name '*.db' | head -1)
python3 "$root/tools/rocpd_stats.py" "$db" "$out/r3_value_forward_kernel_stats.csv" > /dev/null
head -25 "$out/r3_value_forward_kernel_stats.csv" | cut -c1-170
