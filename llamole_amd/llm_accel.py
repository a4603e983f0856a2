"""Optional MI355X acceleration of the untouched HuggingFace LLM's decode step.

The HF model code is not modified; what changes is the kernel underneath ``nn.Linear`` for decode-shaped calls
(<= 64 token rows, bf16, no grad): hipBLASLt picks ~0.7-2.4 TB/s skinny-GEMM kernels there, while the weight-streaming
GEMV / 64x64 LDS-DMA tiles of ``libllamole_hip`` (``ll_linear``) run at 4.7-5.3 TB/s.  Prefill and any other shape fall
through to ``F.linear``.  Everything is enqueued on the caller's current stream, so it composes with the captured
hipGraph of ``llm_decode.GraphedDecoder``.
"""
from __future__ import annotations

import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

MAX_ROWS = 64


def _hip_linear_forward(self: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    K = self.in_features
    if (x.is_cuda and x.dtype == torch.bfloat16 and self.weight.dtype == torch.bfloat16 and not torch.is_grad_enabled()
            and x.numel() // K <= MAX_ROWS and x.shape[-1] == K):
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, N = x2.shape[0], self.out_features
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
        bias = None
        if self.bias is not None:
            bias = getattr(self, "_ll_bias_f32", None)
            if bias is None or bias.device != x.device:
                bias = self.bias.detach().float().contiguous()
                self._ll_bias_f32 = bias
        rc = self._ll_lib.ll_linear(_lib.LL_BF16, x2.data_ptr(), K, self.weight.data_ptr(), K,
                                    bias.data_ptr() if bias is not None else None, out.data_ptr(), N, M, N, K, 0, 0,
                                    torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_linear")
        return out.reshape(*x.shape[:-1], N)
    return F.linear(x, self.weight, self.bias)


def accelerate_linears(model: nn.Module, min_weight_elems: int = 1 << 16) -> int:
    """Route decode-shaped calls of every bf16 ``nn.Linear`` of ``model`` through ``ll_linear``.  Returns the number of
    patched modules.  Weights must stay where they are (the kernel reads ``module.weight`` in place)."""
    lib = _lib.load()
    n = 0
    for mod in model.modules():
        if (type(mod) is nn.Linear and mod.weight.dtype == torch.bfloat16 and mod.weight.is_cuda and mod.weight.is_contiguous()
                and mod.in_features % 8 == 0 and mod.weight.numel() >= min_weight_elems):
            mod._ll_lib = lib
            mod.forward = types.MethodType(_hip_linear_forward, mod)
            n += 1
    return n


def restore_linears(model: nn.Module) -> None:
    for mod in model.modules():
        if type(mod) is nn.Linear and "forward" in mod.__dict__:
            del mod.__dict__["forward"]
