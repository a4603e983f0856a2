#!/bin/bash
# GPU box: if this is one of the boxes where the LDS-DMA ring GEMM is slow (tools/box_check.py), collect the PMC passes of tools/gemm_pmc.sh
# and the clocks for comparison with profiles/r2_gemm_fc1_m512_pmc.txt (a fast box).
root=${GRAFT_REPO_ROOT:-$(pwd)}
line=$(python $root/tools/box_check.py 2>/dev/null | tail -1)
echo "$line"
ring=$(echo "$line" | sed -n 's/.*ring16w \([0-9.]*\) us.*/\1/p')
if python -c "import sys; sys.exit(0 if float('$ring') > 14.0 else 1)"; then
    echo "SLOW BOX" 
    { echo "$line"; rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -v "^=\|^$"; } > $root/gpurun_out/r2_slow_box.txt
    bash $root/tools/gemm_pmc.sh > $root/gpurun_out/r2_gemm_fc1_m512_pmc_slowbox.txt 2>&1
    tail -20 $root/gpurun_out/r2_gemm_fc1_m512_pmc_slowbox.txt
fi
