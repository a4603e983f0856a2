timeout 900 python -m pytest tests/test_parity_full_size_gpu.py tests/test_graphdit_gpu.py tests/test_graphdit_edge_gpu.py -x -q 2>&1 | tail -4
python - <<'PY'
import subprocess, json, os
def run(b, packed):
    code = f"""
import sys, json
from llamole_amd import _lib
lib=_lib.load(); lib.ll_set_m64_packed({packed})
sys.argv=['bench.py','--workload','graphdit','--batch','{b}','--steps','3','--warmup','1','--no-cpu-baseline']
import runpy; runpy.run_path('bench.py', run_name='__main__')
"""
    out = subprocess.run(['python','-c',code],capture_output=True,text=True).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith('{')][-1]); return d['denoise_step_ms']
for b in (1, 1, 1):
    print('B', b, 'row-major:', run(b,0), ' packed:', run(b,1))
PY
