timeout 900 python -m pytest tests/test_parity_full_size_gpu.py tests/test_graphdit_gpu.py tests/test_graphdit_edge_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -4
python - <<'PY'
import subprocess, json, os
def run(b, waves):
    code = f"""
import sys, json
from llamole_amd import _lib
lib=_lib.load(); lib.ll_set_attn_waves({waves})
sys.argv=['bench.py','--workload','graphdit','--batch','{b}','--steps','3','--warmup','1','--no-cpu-baseline']
import runpy; runpy.run_path('bench.py', run_name='__main__')
"""
    out = subprocess.run(['python','-c',code],capture_output=True,text=True,env=dict(os.environ, LL_FUSE_QKV_ATTN='0')).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith('{')][-1]); return d['denoise_step_ms']
for b in (1, 1, 8):
    print('B', b, 'attn waves 2:', run(b,2), ' 4:', run(b,4))
PY
