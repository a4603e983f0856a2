#!/bin/bash
# GPU box, round 3: whole-step counters of the GraphDiT trajectory at batch 1 and 8, and FETCH / WRITE passes over the GIN template head
# (rows16_kernel<plain, f32 out> at [16 x 2048] x [180576 x 2048]^T) folded into gpurun_out/r3_pmc_traffic.json.
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p "$out"
bash "$root/tools/profile_step_pmc.sh" r3_graphdit_b1_step --batch 1
bash "$root/tools/profile_step_pmc.sh" r3_graphdit_b8_step --batch 8
cp "$root/profiles/r2_pmc_traffic.json" "$out/r3_pmc_traffic.json"
for ctr in FETCH_SIZE WRITE_SIZE; do bash "$root/tools/profile_pmc.sh" r3_gin_head $ctr rows16 16 180576 2048 256 0 > /dev/null 2>&1; done
alg=$((180576*2048*2 + 16*2048*2 + 16*180576*4 + 180576*4))
python3 "$root/tools/pmc_traffic.py" gin_head_rows16_m16_n180576_k2048 rows16_kernel "$out/r3_gin_head_FETCH_SIZE.csv" "$out/r3_gin_head_WRITE_SIZE.csv" $alg "$out/r3_pmc_traffic.json" 16 180576 2048
