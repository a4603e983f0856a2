import torch, time
torch.manual_seed(0)
for M in (512, 2048):
    for name,N,K in (("qkv",3072,1024),("proj",1024,1024),("fc1",4096,1024),("fc2",1024,4096)):
        nw = max(2, int(600e6 // (N*K*2)))
        W = torch.randn(nw, N, K, device="cuda", dtype=torch.bfloat16)
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        for lib in ("hipblaslt","hipblas"):
            torch.backends.cuda.preferred_blas_library(lib)
            for i in range(nw): torch.nn.functional.linear(x, W[i])
            torch.cuda.synchronize()
            e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 4*nw
            e0.record()
            for i in range(it): torch.nn.functional.linear(x, W[i % nw])
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1)*1e3/it
            print(f"M={M} {name} {lib}: {us:.1f} us = {2.0*M*N*K/us/1e6:.0f} TF", flush=True)
