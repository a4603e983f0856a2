"""GPU box: prefill (128 tokens, Qwen2-7B shapes) with hipBLASLt vs the LDS-DMA GEMM of libllamole_hip under nn.Linear."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import e2e, llm_accel  # noqa: E402

llm = e2e.build_llm("qwen2-7b", "cuda", torch.bfloat16)
llm_accel.accelerate_linears(llm)
llm_accel.accelerate_elementwise(llm)
ids = torch.randint(5, 150000, (1, 128), device="cuda")


def run(n=5):
    with torch.no_grad():
        for _ in range(2):
            llm(input_ids=ids, logits_to_keep=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            llm(input_ids=ids, logits_to_keep=1)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rows in (64, 128, 256):
    llm_accel.MAX_ROWS = rows
    print(f"MAX_ROWS={rows}: prefill of 128 tokens {run():.2f} ms", flush=True)
