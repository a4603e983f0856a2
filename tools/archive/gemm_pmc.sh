#!/bin/bash
# several PMC passes over the fc1 GEMM at M=512 (dispatch default = cfg 20)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM"; do
  tag=$(echo $ctrs | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_x
  timeout 120 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_x -o x -- python3 $root/tools/gemm_one.py 512 4096 1024 > /tmp/pmc_x.log 2>&1
  f=$(find /tmp/pmc_x -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k=r.get('Kernel_Name','')[:40]; c=r.get('Counter_Name'); v=float(r.get('Counter_Value',0))
    if 'gemm' in k:
        a=agg[(k,c)]; a[0]+=1; a[1]+=v
for (k,c),(n,s) in sorted(agg.items()): print(f"{k} {c} n={n} avg={s/n:.1f}")
PY
  else echo "no csv for $ctrs"; tail -2 /tmp/pmc_x.log; fi
done
