#!/bin/bash
# GPU box: the round-5 bench lines, kernel statistics and per-kernel PMC traffic at HEAD (copied from gpurun_out/ into profiles/ afterwards).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p "$out"
cd "$root"
# ---- kernel traces: the HEADLINE command (no round-4 trace of it existed) and the GraphDiT-only workload at batch 1 / 8
bash tools/profile_bench.sh r5_e2e_b1 --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/profile_bench.sh r5_graphdit_b1_step --workload graphdit --batch 1 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/profile_bench.sh r5_graphdit_b8_step --workload graphdit --steps 3 --warmup 1 > /dev/null 2>&1
# ---- bench lines
python bench.py 2>/dev/null | grep '^{' > "$out/r5_bench_e2e.json"
python bench.py --workload graphdit --steps 3 --warmup 1 2>/dev/null | grep '^{' > "$out/r5_bench_graphdit_b8.json"
python bench.py --workload graphdit --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r5_bench_graphdit_b1.json"
python bench.py --workload graphdit --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r5_bench_graphdit_b16.json"
# the upstream Graph-DiT width (VERDICT r4 next #1c): hidden 1152, 16 heads of 72
python bench.py --workload graphdit --hidden 1152 --depth 28 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r5_bench_graphdit_h1152_b8.json"
python bench.py --workload graphdit --hidden 1152 --depth 28 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r5_bench_graphdit_h1152_b1.json"
python bench.py --workload sft 2>/dev/null | grep '^{' > "$out/r5_bench_sft.json"
python bench.py --workload retro --steps 2 --warmup 1 2>/dev/null | grep '^{' > "$out/r5_bench_retro.json"
for f in r5_bench_e2e r5_bench_graphdit_b8 r5_bench_graphdit_b1 r5_bench_graphdit_b16 r5_bench_graphdit_h1152_b8 r5_bench_graphdit_h1152_b1 r5_bench_sft r5_bench_retro; do python - <<PY
import json
try:
    d = json.loads(open("$out/$f.json").read().strip().splitlines()[-1])
    rd = d.get("roofline_graphdit") or {}
    print("$f", round(d["value"], 3), d["unit"], "ms/step", round(d["ms_per_step"], 2), "dit_step_ms", round(d.get("denoise_step_ms") or 0, 4),
          "roof", round(d["roofline"]["frac"], 3), "roof_dit", round(rd.get("frac", 0) or 0, 3), "live_ms", rd.get("kernel_ms"), "bracket", rd.get("kernel_ms_event_bracketed"),
          "pair", rd.get("event_pair_ms"), "trace", rd.get("kernel_ms_committed_trace"), "b2b", rd.get("kernel_ms_back_to_back"),
          (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
grep -i "gemm_m64_kernel<8, 8, unsigned short, true>\|gemm_bf16_pipeu_kernel<64, 64, 4, 4, 4, unsigned short>" "$out"/r5_graphdit_b1_step_kernel_stats.csv "$out"/r5_graphdit_b8_step_kernel_stats.csv
grep -i "gemv_fused_kernel" "$out"/r5_e2e_b1_kernel_stats.csv | head -5
# ---- PMC traffic per kernel class on single-kernel drivers (tools/gemm_one.py; the whole-trajectory TCC passes crashed rocprofv3 in round 4)
cp "$root/profiles/r3_pmc_traffic.json" "$out/r5_pmc_traffic.json"
pmc() {   # key, kernel substring, algorithmic bytes, M N K, gemm_one args...
    key=$1; sub=$2; alg=$3; M=$4; N=$5; K=$6; shift 6
    for ctr in FETCH_SIZE WRITE_SIZE; do bash tools/profile_pmc.sh r5_$key $ctr "$@" > /dev/null 2>&1; done
    python3 tools/pmc_traffic.py $key "$sub" "$out/r5_${key}_FETCH_SIZE.csv" "$out/r5_${key}_WRITE_SIZE.csv" $alg "$out/r5_pmc_traffic.json" $M $N $K || echo "PMC $key FAILED"
}
pmc llm_gemv_fused_m1_n37888_k3584 gemv_fused_kernel $((37888*3584*2 + 3584*2 + 3584*2 + 18944*2)) 1 37888 3584 fused 1 18944 3584 2 1
pmc fc1_m64 gemm_m64_kernel $((4096*1024*2 + 64*1024*2 + 64*4096*2)) 64 4096 1024 64 4096 1024
pmc fc1_m512 gemm_bf16_pipeu_kernel $((4096*1024*2 + 512*1024*2 + 512*4096*2)) 512 4096 1024 512 4096 1024
pmc llm_rows16_m8_n37888_k3584 rows16_kernel $((37888*3584*2 + 8*3584*2 + 3584*2 + 8*18944*2)) 8 37888 3584 rows16 8 18944 3584 2 1
python3 - <<PY
import json
d = json.load(open("$out/r5_pmc_traffic.json"))
for k, v in d.items():
    print(k, "hbm", round(v["hbm_bytes_per_launch"] / 1e6, 2), "MB  algorithmic", round(v.get("algorithmic_bytes", 0) / 1e6, 2), "MB", "(r5)" if "r5" in v.get("method", "") else "")
PY
