#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE passes (one counter per pass, no trace domains) over the GraphDiT fc1 GEMM at M = 64 / 512 / 2048
# as dispatched at HEAD, folded into profiles-style r2_pmc_traffic.json (bench.py reads it for roofline.traffic).
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out
cp $root/profiles/r1_pmc_traffic.json $out/r2_pmc_traffic.json
for m in 64 512 2048; do
  for ctr in FETCH_SIZE WRITE_SIZE; do bash $root/tools/profile_pmc.sh r2_fc1_m$m $ctr $m 4096 1024 > /dev/null 2>&1; done
  sub=$(python3 - <<PY
import csv
rows=[r["Kernel_Name"] for r in csv.DictReader(open("$out/r2_fc1_m${m}_FETCH_SIZE.csv")) if "gemm" in r["Kernel_Name"]]
print(max(set(rows), key=rows.count)[:70])
PY
)
  alg=$((4096*1024*2 + m*1024*2 + m*4096*2))
  python3 $root/tools/pmc_traffic.py fc1_m$m "$sub" $out/r2_fc1_m${m}_FETCH_SIZE.csv $out/r2_fc1_m${m}_WRITE_SIZE.csv $alg $out/r2_pmc_traffic.json $m 4096 1024
done
