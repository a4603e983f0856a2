import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from llamole_amd import _lib
lib = _lib.load()
def t(M,N,K,cfg,sp):
    ms=C.c_float(); nw=max(2,int(600e6//(N*K*2)))
    rc=lib.ll_gemm_bench(M,N,K,cfg,sp,0,4*nw,nw,C.byref(ms)); return ms.value*1e3 if rc==0 else float('nan')
for M in (512,):
    for name,N,K in (("fc2",1024,4096),("proj",1024,1024),("fc1",4096,1024),("qkv",3072,1024)):
        for sp in (1,2,4,8,16):
            if K % (64*sp) or K//sp < 128: continue
            print(M,name,"splits",sp," ".join(f"{c}:{t(M,N,K,c,sp):.1f}" for c in (20,17,18,19,6,7)), flush=True)
