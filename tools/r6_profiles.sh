#!/bin/bash
# GPU box: the round-6 bench lines, kernel statistics and per-kernel PMC traffic at HEAD (copied from gpurun_out/ into profiles/ afterwards).
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p "$out"
cd "$root"
export LL_PMC_ROUND=r6
# ---- kernel traces: the HEADLINE command and configs[3]'s job on one GPU (64 prompts as one batch, seven-launch layers)
bash tools/profile_bench.sh r6_e2e_b1 --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/profile_bench.sh r6_llama64 --llm llama-3.1-8b --total-prompts 64 --steps 2 --warmup 1 > /dev/null 2>&1
# ---- bench lines
python bench.py 2>/dev/null | grep '^{' > "$out/r6_bench_e2e.json"
python bench.py --llm llama-3.1-8b --total-prompts 64 --steps 4 --warmup 1 2>/dev/null | grep '^{' > "$out/r6_bench_llama_total64_n1.json"
python bench.py --llm llama-3.1-8b --total-prompts 64 --steps 4 --warmup 1 --no-pipeline --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r6_bench_llama_total64_n1_nopipeline.json"
python bench.py --llm qwen2-7b --total-prompts 64 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r6_bench_qwen_total64_n1.json"
python bench.py --llm llama-3.1-8b --total-prompts 64 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r6_bench_llama_total64_batch32.json"
python bench.py --batch 16 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r6_bench_e2e_b16.json"
python bench.py --workload graphdit --steps 3 --warmup 1 2>/dev/null | grep '^{' > "$out/r6_bench_graphdit_b8.json"
python bench.py --workload graphdit --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r6_bench_graphdit_b1.json"
python bench.py --workload graphdit --batch 64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > "$out/r6_bench_graphdit_b64.json"
python bench.py --workload sft 2>/dev/null | grep '^{' > "$out/r6_bench_sft.json"
python bench.py --workload retro --steps 2 --warmup 1 2>/dev/null | grep '^{' > "$out/r6_bench_retro.json"
for f in r6_bench_e2e r6_bench_llama_total64_n1 r6_bench_llama_total64_n1_nopipeline r6_bench_qwen_total64_n1 r6_bench_llama_total64_batch32 r6_bench_e2e_b16 r6_bench_graphdit_b8 r6_bench_graphdit_b1 r6_bench_graphdit_b64 r6_bench_sft r6_bench_retro; do python - <<PY
import json
try:
    d = json.loads(open("$out/$f.json").read().strip().splitlines()[-1])
    rt = d.get("roofline_token") or {}
    print("$f", round(d["value"], 3), d["unit"], "ms/step", round(d["ms_per_step"], 2), "dit_step_ms", round(d.get("denoise_step_ms") or 0, 4),
          "roof", round(d["roofline"]["frac"], 3), "traffic", d["roofline"].get("traffic"), "token", rt.get("frac"), rt.get("token_ms"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
# ---- PMC traffic per kernel class on single-kernel drivers (tools/gemm_one.py).  The file starts EMPTY: bench.py falls back to the older
#      rounds' files, under their own names, for what is not re-measured here (no carried-over entries under a new round's name)
rm -f "$out/r6_pmc_traffic.json"
pmc() {   # key, kernel substring, algorithmic bytes, M N K, gemm_one args...
    key=$1; sub=$2; alg=$3; M=$4; N=$5; K=$6; shift 6
    for ctr in FETCH_SIZE WRITE_SIZE; do bash tools/profile_pmc.sh r6_$key $ctr "$@" > /dev/null 2>&1; done
    python3 tools/pmc_traffic.py $key "$sub" "$out/r6_${key}_FETCH_SIZE.csv" "$out/r6_${key}_WRITE_SIZE.csv" $alg "$out/r6_pmc_traffic.json" $M $N $K || echo "PMC $key FAILED"
}
pmc llm_gemv_fused_m1_n37888_k3584 gemv_fused_kernel $((37888*3584*2 + 3584*2 + 3584*2 + 18944*2)) 1 37888 3584 fused 1 18944 3584 2 1
pmc llm_rows64_m64_n28672_k4096 rows64_kernel $((28672*4096*2 + 64*4096*2 + 64*14336*2)) 64 28672 4096 rows64 64 14336 4096 2 0
pmc llm_rows64_m64_n37888_k3584 rows64_kernel $((37888*3584*2 + 64*3584*2 + 64*18944*2)) 64 37888 3584 rows64 64 18944 3584 2 0
pmc llm_rows64_m32_n28672_k4096 rows64_kernel $((28672*4096*2 + 32*4096*2 + 32*14336*2)) 32 28672 4096 rows64 32 14336 4096 2 0
pmc llm_rows16_m16_n37888_k3584 rows16_kernel $((37888*3584*2 + 16*3584*2 + 3584*2 + 16*18944*2)) 16 37888 3584 rows16 16 18944 3584 2 1
python3 - <<PY
import json
d = json.load(open("$out/r6_pmc_traffic.json"))
for k, v in d.items():
    print(k, "hbm", round(v["hbm_bytes_per_launch"] / 1e6, 2), "MB  algorithmic", round(v.get("algorithmic_bytes", 0) / 1e6, 2), "MB", round(v["hbm_bytes_per_launch"] / v["algorithmic_bytes"], 4))
PY
