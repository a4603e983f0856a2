#!/bin/bash
# GPU box: SQ / LDS counters of rows64_kernel on the gate|up shape at 32 and 64 rows, one rocprofv3 --pmc pass per counter group.
# usage: tools/rows64_pmc.sh [N K EPI]
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r6; mkdir -p $out
N=${1:-14336}; K=${2:-4096}; EPI=${3:-2}
cd /tmp && export TMPDIR=/tmp
for M in 32 64; do
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    tag=r64_m${M}_$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/pmc_$tag
    timeout 180 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$tag -o $tag -- python3 $root/tools/gemm_one.py rows64 $M $N $K $EPI 0 > /tmp/pmc_$tag.log 2>&1
    f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 - "$f" "M=$M" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "rows64_kernel" not in r["Kernel_Name"]:
        continue
    a = agg[r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, (n, v) in agg.items():
    print(sys.argv[2], k, "dispatches", n, "avg", round(v / n, 1))
PY
  done
done 2>&1 | tee $out/rows64_pmc.txt
