timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 12 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r2_bench_e2e.json; cat gpurun_out/r2_bench_e2e.json | cut -c1-400
