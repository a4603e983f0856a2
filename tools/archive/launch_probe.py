import ctypes as C, sys
sys.path.insert(0, ".")
from llamole_amd import _lib
lib = _lib.load()
def probe(kind, graph=1, n=2000):
    us = C.c_float()
    _lib.check(lib.ll_launch_bench(kind, n, graph, C.byref(us)))
    return us.value
for kind, name in enumerate(["empty", "load+store", "dependent loads"]):
    print(f"{name:16s} eager {probe(kind,0):.2f}  graph {probe(kind,1):.2f} us/kernel")
for blocks in (1, 4, 16, 64, 128, 256, 512, 960):
    print(f"blocks={blocks:4d} x256thr float4: copy {probe(100*blocks+3):.2f}  read {probe(100*blocks+4):.2f}  write {probe(100*blocks+5):.2f} us")
for kind, name in enumerate(["empty kernel", "2-argument kernel", "linear_launch -> gemm_m64_kernel", "linear_launch -> LDS-DMA ring"]):
    us = C.c_float()
    _lib.check(lib.ll_host_launch_probe(kind, 6000, C.byref(us)))
    print(f"host time per enqueued launch, {name:34s} {us.value:.2f} us")
