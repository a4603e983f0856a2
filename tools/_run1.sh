hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -DLL_QA_PROBE -Wno-unused-function tools/qkv_attn_probe.hip -o /tmp/qkv_attn_probe 2>&1 | grep -v warning | head
for b in 8; do /tmp/qkv_attn_probe $b | head -2; done
timeout 900 python -m pytest tests/test_parity_full_size_gpu.py -x -q -k "fused or ref_default or attn_mfma" 2>&1 | tail -5
run() { python bench.py --workload graphdit --batch $1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d.get('denoise_step_ms'))"; }
for b in 2 8; do
LL_FUSE_QKV_ATTN=0 run $b "B=$b unfused"
LL_FUSE_QKV_ATTN=1 run $b "B=$b fused"
done
