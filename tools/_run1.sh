timeout 900 python -m pytest tests/test_graphdit_gpu.py tests/test_graphdit_edge_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -3
run() { python bench.py --workload graphdit --batch $1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d.get('denoise_step_ms'))"; }
for b in 1 8; do
LL_STEPS_PER_GRAPH=1 run $b "B=$b spg1"
LL_STEPS_PER_GRAPH=5 run $b "B=$b spg5"
LL_STEPS_PER_GRAPH=10 run $b "B=$b spg10"
LL_STEPS_PER_GRAPH=1 run $b "B=$b spg1"
LL_STEPS_PER_GRAPH=5 run $b "B=$b spg5"
done
rune() { python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
LL_STEPS_PER_GRAPH=1 rune "e2e spg1"
LL_STEPS_PER_GRAPH=5 rune "e2e spg5"
LL_STEPS_PER_GRAPH=1 rune "e2e spg1"
LL_STEPS_PER_GRAPH=5 rune "e2e spg5"
