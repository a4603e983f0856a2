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


# ------------------------------------------------------------------------------------------ fused elementwise ops
def _decode_shaped(x: torch.Tensor, width: int) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and x.shape[-1] == width
            and x.numel() // width <= MAX_ROWS)


def _rmsnorm_forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
    H = self.weight.shape[0]
    if _decode_shaped(hidden_states, H) and self.weight.dtype == torch.bfloat16 and H % 8 == 0 and H <= 8192:
        x = hidden_states.reshape(-1, H)
        if not x.is_contiguous():
            x = x.contiguous()
        out = torch.empty_like(x)
        rc = self._ll_lib.ll_rmsnorm_bf16(x.data_ptr(), self.weight.data_ptr(), out.data_ptr(), x.shape[0], H,
                                          float(self.variance_epsilon), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_rmsnorm_bf16")
        return out.reshape(hidden_states.shape)
    return self._ll_orig_forward(hidden_states)


def _mlp_forward(self, x: torch.Tensor) -> torch.Tensor:
    if _decode_shaped(x, self.gate_proj.in_features) and self.gate_proj.out_features % 8 == 0:
        g, u = self.gate_proj(x), self.up_proj(x)
        if g.dtype == torch.bfloat16 and g.is_contiguous() and u.is_contiguous():
            h = torch.empty_like(g)
            rc = self._ll_lib.ll_silu_mul_bf16(g.data_ptr(), u.data_ptr(), h.data_ptr(), g.numel(),
                                               torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                _lib.check(rc, "ll_silu_mul_bf16")
            return self.down_proj(h)
        return self.down_proj(self.act_fn(g) * u)
    return self._ll_orig_forward(x)


def _make_rope(orig, lib):
    import ctypes as C
    I64x3 = C.c_int64 * 3

    def apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=1, **kw):
        if (unsqueeze_dim == 1 and q.dim() == 4 and q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16
                and cos.dtype == torch.bfloat16 and cos.dim() == 3 and not torch.is_grad_enabled() and q.shape[2] <= MAX_ROWS
                and q.stride(3) == 1 and k.stride(3) == 1 and cos.stride(2) == 1 and sin.stride() == cos.stride()
                and q.shape[3] % 2 == 0 and q.shape[3] <= 128 and cos.shape[-1] == q.shape[3]):
            B, nh, S, D = q.shape
            nkv = k.shape[1]
            qo = torch.empty((B, nh, S, D), dtype=q.dtype, device=q.device)
            ko = torch.empty((B, nkv, S, D), dtype=k.dtype, device=k.device)
            cs0 = 0 if cos.shape[0] == 1 else cos.stride(0)
            rc = lib.ll_rope_bf16(q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), qo.data_ptr(), ko.data_ptr(),
                                  B, nh, nkv, S, D, I64x3(q.stride(0), q.stride(1), q.stride(2)),
                                  I64x3(k.stride(0), k.stride(1), k.stride(2)), I64x3(cs0, cos.stride(1), 1),
                                  torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                _lib.check(rc, "ll_rope_bf16")
            return qo, ko
        return orig(q, k, cos, sin, unsqueeze_dim=unsqueeze_dim, **kw)

    apply_rotary_pos_emb._ll_orig = orig
    return apply_rotary_pos_emb


def _is_silu(fn) -> bool:
    return fn is F.silu or type(fn).__name__ in ("SiLU", "SiLUActivation")


def accelerate_elementwise(model: nn.Module) -> dict:
    """Fuse the decode-step RMSNorm / rotary embedding / SiLU*mul of a Llama-family HF model into single launches
    (ll_rmsnorm_bf16, ll_rope_bf16, ll_silu_mul_bf16).  Non-decode shapes keep the original HF code path."""
    import sys
    lib = _lib.load()
    n_norm = n_mlp = 0
    for mod in model.modules():
        cls = type(mod).__name__
        if cls.endswith("RMSNorm") and hasattr(mod, "variance_epsilon") and hasattr(mod, "weight") and "forward" not in mod.__dict__:
            mod._ll_lib, mod._ll_orig_forward = lib, mod.forward
            mod.forward = types.MethodType(_rmsnorm_forward, mod)
            n_norm += 1
        elif (all(hasattr(mod, a) for a in ("gate_proj", "up_proj", "down_proj", "act_fn")) and _is_silu(mod.act_fn)
              and "forward" not in mod.__dict__):
            mod._ll_lib, mod._ll_orig_forward = lib, mod.forward
            mod.forward = types.MethodType(_mlp_forward, mod)
            n_mlp += 1
    rope = 0
    base = getattr(model, "model", model)
    m = sys.modules.get(type(base).__module__)
    if m is not None and hasattr(m, "apply_rotary_pos_emb") and not hasattr(m.apply_rotary_pos_emb, "_ll_orig"):
        m.apply_rotary_pos_emb = _make_rope(m.apply_rotary_pos_emb, lib)
        rope = 1
    return {"rmsnorm": n_norm, "mlp": n_mlp, "rope": rope}


def restore_elementwise(model: nn.Module) -> None:
    import sys
    for mod in model.modules():
        if "_ll_orig_forward" in mod.__dict__ and "forward" in mod.__dict__:
            del mod.__dict__["forward"]
            del mod.__dict__["_ll_orig_forward"]
    base = getattr(model, "model", model)
    m = sys.modules.get(type(base).__module__)
    if m is not None and hasattr(getattr(m, "apply_rotary_pos_emb", None), "_ll_orig"):
        m.apply_rotary_pos_emb = m.apply_rotary_pos_emb._ll_orig
