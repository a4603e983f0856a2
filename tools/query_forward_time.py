"""Device time of the query-token forward (GraphedDecoder.continue_hidden: 9 tokens on top of the decode's KV cache, replayed hipGraph) on
the five-launch layers and op by op (LLAMOLE_FUSED_SUFFIX=0).  python tools/query_forward_time.py [qwen2-7b|llama-3.1-8b]"""
import os
import sys

os.environ.setdefault("LLAMOLE_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llamole_amd import e2e  # noqa: E402
from llamole_amd.llm_accel import accelerate_llm  # noqa: E402
from llamole_amd.llm_decode import GraphedDecoder  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "qwen2-7b"
llm = e2e.build_llm(name, "cuda", torch.bfloat16)
info = accelerate_llm(llm)
g = torch.Generator().manual_seed(0)
P, new = 48, 128
prompt = torch.randint(5, 100000, (1, P), generator=g).cuda()
mask = torch.ones_like(prompt)
tail = torch.randint(5, 100000, (1, 9), generator=g).cuda()
for mode in ("1", "0", "1", "0"):
    os.environ["LLAMOLE_FUSED_SUFFIX"] = mode
    d = GraphedDecoder(llm, use_graph=True, fused_cache=bool(info.get("decode_attention")))
    d.generate(prompt, mask, max_new_tokens=new, do_sample=False, pad_token_id=0, eos_token_id=[])
    for _ in range(3):
        d.continue_hidden(tail, P + new - 9)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        d.continue_hidden(tail, P + new - 9)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: query forward, {'five-launch layers' if mode == '1' else 'op by op'}: {e0.elapsed_time(e1) / 20:.3f} ms per call (replayed graph)")
