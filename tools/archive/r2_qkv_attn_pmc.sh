#!/bin/bash
# GPU box: PMC passes (one counter group per pass, no trace domains) over the stand-alone probe of qkv_attn_kernel at batch 8
# -> gpurun_out/r2_qkv_attn_pmc.txt (per-launch averages of the kernel's dispatches)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r2_qkv_attn_pmc.txt
flags="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -DLL_QA_PROBE -Wno-unused-function"
hipcc $flags $root/tools/qkv_attn_probe.hip -o /tmp/qkv_attn_probe 2>/dev/null
cd /tmp && export TMPDIR=/tmp
: > $out
for ctr in FETCH_SIZE WRITE_SIZE MfmaUtil "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $ctr | tr ' ' '_')
  rm -rf /tmp/pmc_qa_$tag
  timeout 120 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_qa_$tag -o qa -- /tmp/qkv_attn_probe 8 > /tmp/pmc_qa_$tag.log 2>&1
  f=$(find /tmp/pmc_qa_$tag -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "qkv_attn_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k}: mean per launch {sum(v) / len(v):.6g} over {len(v)} launches")
PY
  else echo "$ctr: pass failed" >> $out; fi
done
cat $out
