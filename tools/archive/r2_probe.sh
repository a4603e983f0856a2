#!/bin/bash
# GPU box: tools/qkv_attn_probe.hip in its production form at batch 1..32 and in three variants at batch 8 -> gpurun_out/r2_qkv_attn_probe.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r2_qkv_attn_probe.txt
flags="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -DLL_QA_PROBE -Wno-unused-function"
: > $out
hipcc $flags $root/tools/qkv_attn_probe.hip -o /tmp/qkv_attn_probe 2>/dev/null
for b in 1 2 4 8 16 32; do /tmp/qkv_attn_probe $b >> $out; done
for m in 1 3 5; do
    hipcc $flags -DLL_QA_MODE=$m $root/tools/qkv_attn_probe.hip -o /tmp/qkv_attn_probe_m 2>/dev/null
    echo "--- variant LL_QA_MODE=$m (1: weight blocks requested after the panel is staged; 3: every wave's panel pieces queued before any weight block; 5: K loop without MFMA / LDS reads)" >> $out
    /tmp/qkv_attn_probe_m 8 >> $out
done
cat $out
