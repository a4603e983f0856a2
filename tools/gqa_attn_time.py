"""Device time of ll_decode_attn_rope_bf16 (rope + KV append + grouped-query decode attention) for B sequences, distinct caches per call
(a model's layers).  python tools/gqa_attn_time.py [B ...]   env: NH NKV D MAXLEN POS"""
import os
import sys

os.environ.setdefault("LLAMOLE_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llamole_amd import _lib  # noqa: E402

lib = _lib.load()
nh, nkv, D = int(os.environ.get("NH", 32)), int(os.environ.get("NKV", 8)), int(os.environ.get("D", 128))
maxlen, p0 = int(os.environ.get("MAXLEN", 256)), int(os.environ.get("POS", 191))
st = torch.cuda.current_stream().cuda_stream
for B in [int(a) for a in sys.argv[1:]] or [16, 32, 64]:
    nl = max(2, int(600e6 // (2 * B * nkv * maxlen * D * 2)) + 1)
    Ks = [torch.randn(B, nkv, maxlen, D, device="cuda", dtype=torch.bfloat16) for _ in range(nl)]
    Vs = [torch.randn(B, nkv, maxlen, D, device="cuda", dtype=torch.bfloat16) for _ in range(nl)]
    qkv = torch.randn(B, (nh + 2 * nkv) * D, device="cuda", dtype=torch.bfloat16)
    cos = torch.randn(B, D, device="cuda", dtype=torch.bfloat16)
    sin = torch.randn(B, D, device="cuda", dtype=torch.bfloat16)
    pos = torch.tensor([p0], dtype=torch.long, device="cuda")
    mask = (torch.arange(maxlen, device="cuda")[None, :] <= p0).expand(B, maxlen).contiguous()
    out = torch.empty(B, nh * D, device="cuda", dtype=torch.bfloat16)

    def run(i):
        rc = lib.ll_decode_attn_rope_bf16(qkv.data_ptr(), qkv.stride(0), cos.data_ptr(), sin.data_ptr(), D, Ks[i % nl].data_ptr(), Vs[i % nl].data_ptr(),
                                          pos.data_ptr(), mask.data_ptr(), maxlen, out.data_ptr(), B, nh, nkv, maxlen, D, D ** -0.5, st)
        assert rc == 0

    for i in range(5):
        run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    n = 60
    e0.record()
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    mb = 2 * B * nkv * (p0 + 1) * D * 2 / 1e6
    print(f"B={B} nh={nh} nkv={nkv} D={D} maxlen={maxlen} pos={p0}: {us:6.1f} us per launch, {mb:.1f} MB of keys + values -> {mb / us / 1e3 * 1e3:.2f} TB/s")
