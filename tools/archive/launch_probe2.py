import ctypes as C, sys, os
sys.path.insert(0, ".")
import torch
from llamole_amd import _lib
lib = _lib.load()
def probe(kind, graph=1, n=1000):
    us = C.c_float()
    _lib.check(lib.ll_launch_bench(kind, n, graph, C.byref(us)))
    return us.value
a = torch.zeros(1 << 28, dtype=torch.uint8, device="cuda").view(-1); b = torch.zeros(1 << 22, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
lib.ll_launch_bench_set_buffers(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()))
for blocks in (16, 64, 120):
    print(f"blocks={blocks:4d}x256: contiguous {probe(100*blocks+4):.2f}  same-4KB {probe(100*blocks+6):.2f}  stride-64KB {probe(100*blocks+8):.2f}  stride-2MB {probe(100*blocks+7):.2f} us")
