"""Probe: rope + append + attention and the o_proj GEMV behind it as ONE launch (ll_decode_attn_oproj_probe: 28 head workgroups signal, 224 resident
o_proj workgroups hold their first weight block and wait) against the two launches of the product, inside the full five-launch decoder layer
at Qwen2-7B shapes, rotating over 8 weight sets (HBM-resident stream), hipGraph of 24 layers.  Bit-identity of every output first.
usage: python tools/attn_oproj_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib  # noqa: E402

lib = _lib.load()
H, I, nh, nkv, D, L = 3584, 18944, 28, 4, 128, 256
nq, nqkv = nh * D, (nh + 2 * nkv) * D
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)      # noqa: E731
sets = []
for _ in range(8):
    t = dict(wqkv=r(nqkv, H, sc=H ** -0.5).bfloat16(), bqkv=r(nqkv, sc=0.1).float(), n1=(1 + 0.1 * r(H)).bfloat16(), wo=r(H, nq, sc=nq ** -0.5).bfloat16(),
             wgu=r(2 * I, H, sc=H ** -0.5).bfloat16(), n2=(1 + 0.1 * r(H)).bfloat16(), wd=r(H, I, sc=I ** -0.5).bfloat16(),
             K=r(1, nkv, L, D).bfloat16(), V=r(1, nkv, L, D).bfloat16())
    sets.append({k: v.cuda() for k, v in t.items()})
h0 = r(1, H).bfloat16().cuda()
cos, sin = r(1, 1, D).bfloat16().cuda(), r(1, 1, D).bfloat16().cuda()
pos = torch.tensor([200], dtype=torch.int64, device="cuda")
mask = (torch.arange(L, device="cuda") <= 200).view(1, 1, 1, L).contiguous()
mk = lambda n: torch.zeros(1, n, dtype=torch.bfloat16, device="cuda")      # noqa: E731
buf = {v: dict(qkv=mk(nqkv), att=mk(nq), h1=mk(H), act=mk(I), h2=mk(H)) for v in ("two", "one")}
ctr = torch.zeros(1024, dtype=torch.int32, device="cuda")
SLEEP = int(os.environ.get("PROBE_SLEEP", "1"))


def layer(t, x, b, fused):
    s = torch.cuda.current_stream().cuda_stream
    ck = _lib.check
    ck(lib.ll_gemv_fused_bf16(x.data_ptr(), H, t["wqkv"].data_ptr(), H, t["bqkv"].data_ptr(), t["n1"].data_ptr(), 1e-6, None, 0, b["qkv"].data_ptr(), nqkv,
                              1, nqkv, H, 0, s), "qkv")
    if fused:
        ck(lib.ll_decode_attn_oproj_probe(b["qkv"].data_ptr(), nqkv, cos.data_ptr(), sin.data_ptr(), t["K"].data_ptr(), t["V"].data_ptr(), pos.data_ptr(),
                                          mask.data_ptr(), b["att"].data_ptr(), nh, nkv, L, D, D ** -0.5, t["wo"].data_ptr(), None, x.data_ptr(),
                                          b["h1"].data_ptr(), H, ctr.data_ptr(), SLEEP, s), "attn+o_proj")
    else:
        ck(lib.ll_decode_attn_rope_bf16(b["qkv"].data_ptr(), nqkv, cos.data_ptr(), sin.data_ptr(), 0, t["K"].data_ptr(), t["V"].data_ptr(), pos.data_ptr(),
                                        mask.data_ptr(), 0, b["att"].data_ptr(), 1, nh, nkv, L, D, D ** -0.5, s), "attn")
        ck(lib.ll_gemv_fused_bf16(b["att"].data_ptr(), nq, t["wo"].data_ptr(), nq, None, None, 0.0, x.data_ptr(), H, b["h1"].data_ptr(), H, 1, H, nq, 1, s), "o_proj")
    ck(lib.ll_gemv_fused_bf16(b["h1"].data_ptr(), H, t["wgu"].data_ptr(), H, None, t["n2"].data_ptr(), 1e-6, None, 0, b["act"].data_ptr(), I, 1, I, H, 2, s), "gate|up")
    ck(lib.ll_gemv_fused_bf16(b["act"].data_ptr(), I, t["wd"].data_ptr(), I, None, None, 0.0, b["h1"].data_ptr(), H, b["h2"].data_ptr(), H, 1, H, I, 1, s), "down")
    return b["h2"]


# ---- bit-identity: 40 layers in a row (each layer's h2 is the next one's input), K/V caches restored in between
kv0 = [(t["K"].clone(), t["V"].clone()) for t in sets]
outs = {}
for name, fused in (("two", False), ("one", True)):
    for t, (k, v) in zip(sets, kv0):
        t["K"].copy_(k), t["V"].copy_(v)
    x = h0.clone()
    for i in range(40):
        x = layer(sets[i % 8], x, buf[name], fused).clone()
    torch.cuda.synchronize()
    outs[name] = (x, buf[name]["att"].clone(), buf[name]["h1"].clone(), [t["K"].clone() for t in sets])
same = all(torch.equal(a, b) for a, b in zip(outs["two"][:3], outs["one"][:3])) and all(torch.equal(a, b) for a, b in zip(outs["two"][3], outs["one"][3]))
print("bit-identical after 40 chained layers:", same, " error word:", int(ctr[256]), " counters zeroed:", int(ctr[:300].abs().sum()) == 0, flush=True)


def graphed(fused):
    b = buf["one" if fused else "two"]
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(8):
            layer(sets[i % 8], h0, b, fused)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            x = h0
            for i in range(24):
                x = layer(sets[i % 8], x, b, fused)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (10 * 24) * 1e3


for rep in range(3):
    print(f"five launches per layer: {graphed(False):7.2f} us   attention + o_proj as one (sleep {SLEEP}): {graphed(True):7.2f} us", flush=True)
print("error word:", int(ctr[256]))
