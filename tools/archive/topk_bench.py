#!/usr/bin/env python3
"""softmax + top-k over the template logits (ll_softmax_topk): one workgroup per row vs the chunked two-stage form."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamole_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    out = {}
    for rows, D, k in [(16, 180576, 50), (1, 180576, 50), (64, 180576, 50)]:
        logits = (torch.randn(rows, D, device="cuda") * 3).contiguous()
        probs = torch.empty(rows, k, device="cuda")
        idx = torch.empty(rows, k, device="cuda", dtype=torch.int32)
        for single in (1, 0):
            lib.ll_set_topk_single(single)
            st = _lib.current_stream_ptr()
            for _ in range(5):
                lib.ll_softmax_topk(_lib.dptr(logits), rows, D, k, _lib.dptr(probs), _lib.dptr(idx), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                lib.ll_softmax_topk(_lib.dptr(logits), rows, D, k, _lib.dptr(probs), _lib.dptr(idx), st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
            out[f"rows{rows}_{'single' if single else 'chunked'}_us"] = us
            out[f"rows{rows}_{'single' if single else 'chunked'}_GBs"] = rows * D * 4 / us / 1e3
        lib.ll_set_topk_single(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
