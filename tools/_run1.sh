python - <<'PY'
import ctypes as C
from llamole_amd import _lib
lib=_lib.load()
def t(M,N,K,cfg,sp,nw):
    ms=C.c_float()
    rc=lib.ll_gemm_bench(M,N,K,cfg,sp,0,160,nw,C.byref(ms)); return ms.value*1e3 if rc==0 else float('nan')
for M in (64,512):
  for nw in (40,1):
    print(f"M={M} nweights={nw}: fc1 dispatch {t(M,4096,1024,-1,1,nw):.1f} xw {t(M,4096,1024,-2,1,nw):.1f} | fc2 dispatch(s2) {t(M,1024,4096,-1,2,nw):.1f} (s4) {t(M,1024,4096,-1,4,nw):.1f} xw(s4) {t(M,1024,4096,-2,4,nw):.1f} | qkv dispatch {t(M,3072,1024,-1,1,nw):.1f} xw {t(M,3072,1024,-2,1,nw):.1f} | proj dispatch(s2) {t(M,1024,1024,-1,2,nw):.1f} xw {t(M,1024,1024,-2,1,nw):.1f}")
PY
run() { python bench.py --workload graphdit --batch $1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d.get('denoise_step_ms'))"; }
for b in 8; do
LL_XW_GEMM=0 run $b "B=$b ring"
LL_XW_GEMM=1 run $b "B=$b xw fc1"
LL_XW_GEMM=0 run $b "B=$b ring"
LL_XW_GEMM=1 run $b "B=$b xw fc1"
done
