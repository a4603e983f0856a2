#!/usr/bin/env python3
"""One A* value-estimate forward of the retro workload alone (Qwen2-7B architecture, 256 prompts of ~130 tokens): wall time and -- under
rocprofv3 --kernel-trace --stats -- where its GPU time goes (vendor GEMMs vs attention vs elementwise)."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import e2e  # noqa: E402
from llamole_amd.planner import ReactionView  # noqa: E402

dev = torch.device("cuda")
llm = e2e.build_llm(sys.argv[1] if len(sys.argv) > 1 else "qwen2-7b", dev, torch.bfloat16)
orch, tok = e2e.build_orchestrator(llm, types.SimpleNamespace(text_input_size=768, check_valid=lambda s: True), dev)
orch.enable_mi355x_decode()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
items = [(f"M{i}", ReactionView(2, f"T{i * 7}", [f"M{i}", f"M{i + 500}"])) for i in range(n)]
L = len(tok.encode(tok.apply_chat_template([{"role": "user", "content": orch._complexity_prompt(*items[0])}], tokenize=False, add_generation_prompt=True)))
params = sum(p.numel() for n_, p in llm.named_parameters() if "embed_tokens" not in n_ and "lm_head" not in n_)
only_default = len(sys.argv) > 3 and sys.argv[3] == "default"      # under the profiler: the product's configuration alone
for pmin in ((type(orch).value_prefix_min,) if only_default else (type(orch).value_prefix_min, 0)):      # the prompts' shared opening served from one set of keys / values | every prompt forwarded whole
    orch.value_prefix_min = pmin
    for _ in range(2):
        orch.estimate_synthesis_complexity_batch(items, None, 0, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = orch.estimate_synthesis_complexity_batch(items, None, 0, 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    P = orch.last_value_opening
    print(f"{n} prompts x {L} tokens, {P} opening tokens shared: {dt * 1e3:.1f} ms per call, "
          f"{2 * params * n * (L - P) / dt / 1e12:.0f} TFLOP/s over the decoder stack, values {out[:2]}")
