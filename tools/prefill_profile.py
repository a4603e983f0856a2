"""The prompt prefill alone (Qwen2-7B shapes, 128 tokens, the accelerated stack of bench.py): `iters` generate() calls of ONE new token each.
Run under rocprofv3 --kernel-trace for the per-kernel breakdown (tools/rocpd_stats.py); prints the HIP-event time per prefill.
usage: python tools/prefill_profile.py [tokens] [iters]"""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import e2e                       # noqa: E402
from llamole_amd.llm_accel import accelerate_llm  # noqa: E402
from llamole_amd.llm_decode import GraphedDecoder  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
llm = e2e.build_llm("qwen2-7b", "cuda", torch.bfloat16)
info = accelerate_llm(llm)
dec = GraphedDecoder(llm, use_graph=True, fused_cache=bool(info.get("decode_attention")))
g = torch.Generator().manual_seed(0)
prompt = torch.randint(5, 150000, (1, P), generator=g).cuda()
mask = torch.ones_like(prompt)
kw = dict(max_new_tokens=1, do_sample=False, pad_token_id=0, eos_token_id=[])
for _ in range(3):
    dec.generate(prompt, mask, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    dec.generate(prompt, mask, **kw)
e1.record()
torch.cuda.synchronize()
print(f"prefill of {P} tokens: {e0.elapsed_time(e1) / iters:.3f} ms per call (HIP events, host enqueue included)")
