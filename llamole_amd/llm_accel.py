"""Optional MI355X acceleration of the untouched HuggingFace LLM's decode step.

The HF model code is not modified; what changes is the kernel underneath its modules for decode-shaped calls (bf16, no grad):
  * 1-2 sequences : five launches per decoder layer on the f32-FMA GEMVs (RMSNorm prologue, residual / SiLU*mul epilogues; bit-identical
                    to HF's op order), rope + KV append + attention in one launch, one-launch prologue and sampler;
  * 3-16 sequences: the same five launches on the weight-streaming MFMA Linear (``ll_linear_rows16_bf16``);
  * 17-64 sequences (round 6): seven launches per layer on copies of the weights in MFMA operand order (``ll_linear_rows64_bf16``), the
                    RMSNorm between two Linears split over producer and consumer, grouped-query attention with one workgroup per
                    (KV head, sequence) -- the batch the reference's DataLoader hands ``language_model.generate`` (eval/workflow.py:89-91);
  * anything else (prefill, larger batches, a dynamic cache) falls through to the HF code, with ``ll_linear`` under ``nn.Linear`` up to 128 rows.
Everything is enqueued on the caller's current stream, so it composes with the captured hipGraph of ``llm_decode.GraphedDecoder``.
"""
from __future__ import annotations

import os
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

MAX_ROWS = 128           # GEMV / skinny-GEMM path under nn.Linear (128-token prefill: 8.7 ms vs 10.0 ms on hipBLASLt, tools/prefill_probe.py)
MAX_ROWS16 = 16         # ll_linear_rows16_bf16: one MFMA column block of token rows (batched decode: 3..16 sequences)
MAX_STREAM_ROWS = 64    # ll_linear_rows64_bf16: up to four column blocks per weight fragment (17..64 sequences per GPU, BASELINE configs[3])
# up to here the f32-FMA GEMVs run (RMSNorm prologue fused, bit-identical to the op-by-op path); from the next row count on the MFMA
# stream does (tools/batch_sweep.sh: batch 4 = 603 ms per step on the FMA path vs 467 ms, batch 2 = 456 vs 439 ms -- the default
# stays at 2 so that two-sequence decode keeps the bit-identity with HF's op order; LLAMOLE_FMA_GEMV_ROWS=1 trades it for the 4 %)
FMA_GEMV_ROWS = int(os.environ.get("LLAMOLE_FMA_GEMV_ROWS", "2"))
PACK64_MIN_N = 32768    # nn.Linear modules at least this wide (lm_head) keep a packed copy for 17..64 rows
MAX_APPEND_ROWS = 16     # new positions per call served by the fused KV append / decode attention (decode, query tail)
# row-parallel elementwise kernels (RMSNorm, rotary, SiLU*mul) also serve prefill-sized calls.  Round 3: the A* value estimates push
# 256 prompts x ~144 tokens = 37 k rows through the model per forward; above the old 16 k-row limit HF's op-by-op RMSNorm (pow, mean, rsqrt,
# two casts, two multiplies: ~0.7 ms per norm at 37 k x 3584) and act_fn(gate) * up (1.3 ms per layer) were 17 % of that forward
MAX_EW_ROWS = 1 << 20
# vocabulary-sized outputs for a few hundred rows (the lm_head of a batched `logits_to_keep = 1` forward: 256 x 152 064 x 3584): hipBLASLt
# picks a 256 x 16 tile there and takes 32 ms (8.6 TFLOP/s); the LDS-DMA ring GEMM streams the 1.09 GB of weights once
MAX_WIDE_ROWS, WIDE_N = 1024, 65536
PREFILL_SPLITK = os.environ.get("LLAMOLE_PREFILL_SPLITK", "1") != "0"


def _versions(*tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors if t is not None)


class _Packed64:
    """A weight in MFMA operand order for ll_linear_rows64_bf16 (made once by ll_rows64_pack_bf16; the same size as the weight).  The copy
    follows its source: ``sync`` re-packs IN PLACE when (data_ptr, _version) of the sources changed (captured graphs hold its address)."""

    def __init__(self, lib, w: torch.Tensor, key):
        self.rows, self.K = w.shape
        n = int(lib.ll_rows64_packed_elems(self.rows, self.K))
        if n < 0:
            raise ValueError(f"ll_rows64_pack_bf16: K={self.K} must be a multiple of 32")
        self.t = torch.empty(n, dtype=torch.bfloat16, device=w.device)
        self.key = None
        self.sync(lib, w, key)

    def sync(self, lib, w: torch.Tensor, key) -> bool:
        if self.key == key:
            return False
        assert w.shape == (self.rows, self.K) and w.stride(1) == 1
        rc = lib.ll_rows64_pack_bf16(w.data_ptr(), w.stride(0), self.rows, self.K, self.t.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_rows64_pack_bf16")
        self.key = key
        return True


def _rows64(lib, x2, ldx, wp: "_Packed64", bias, res, N, epi, row_ssq=None, eps=0.0, next_norm=None):
    """17..64 token rows through ll_linear_rows64_bf16 on a packed weight; matrices with few 64-row groups (o_proj, down_proj) get scratch
    for the cross-workgroup K split (f32 slabs summed in slice order by a second launch on the same stream).
    ``row_ssq`` [M, chunks] f32: x2 is a pre-scaled row bf16(h * w_norm) and the accumulator is multiplied by rsqrt(mean(h^2) + eps).
    ``next_norm``: also return (bf16(out * next_norm), per-chunk sums of out^2) -- the two halves of the NEXT Linear's RMSNorm."""
    M, K = x2.shape[0], wp.K
    out = torch.empty(M, N, dtype=torch.bfloat16, device=x2.device)
    xs = ssq = None
    if next_norm is not None:
        xs = torch.empty(M, N, dtype=torch.bfloat16, device=x2.device)
        ssq = torch.empty(M, int(lib.ll_rows64_ssq_chunks(N)), dtype=torch.float32, device=x2.device)
    ws, wsb = None, 0
    if epi != 2 and (next_norm is not None or N < 192 * 64):
        wsb = int(lib.ll_linear_rows64_workspace_bytes(M, N))
        ws = torch.empty(wsb, dtype=torch.uint8, device=x2.device)
    rc = lib.ll_linear_rows64_bf16(x2.data_ptr(), ldx, wp.t.data_ptr(), bias.data_ptr() if bias is not None else None,
                                   res.data_ptr() if res is not None else None, res.stride(0) if res is not None else 0,
                                   out.data_ptr(), N, M, N, K, epi, row_ssq.data_ptr() if row_ssq is not None else None,
                                   row_ssq.shape[1] if row_ssq is not None else 0, eps,
                                   next_norm.data_ptr() if next_norm is not None else None, xs.data_ptr() if xs is not None else None, N,
                                   ssq.data_ptr() if ssq is not None else None, ws.data_ptr() if ws is not None else None, wsb,
                                   torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        _lib.check(rc, "ll_linear_rows64_bf16")
    return out if next_norm is None else (out, xs, ssq)


def _prenorm64(lib, x2, norm_w):
    """(bf16(x * norm_w), per-chunk sums of x^2) of rows that no ll_linear_rows64_bf16 call produced (the embedding rows)."""
    M, N = x2.shape
    xs = torch.empty(M, N, dtype=torch.bfloat16, device=x2.device)
    ssq = torch.empty(M, int(lib.ll_rows64_ssq_chunks(N)), dtype=torch.float32, device=x2.device)
    rc = lib.ll_rows64_prenorm_bf16(x2.data_ptr(), x2.stride(0), norm_w.data_ptr(), xs.data_ptr(), N, ssq.data_ptr(), M, N,
                                    torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        _lib.check(rc, "ll_rows64_prenorm_bf16")
    return xs, ssq


def _hip_linear_forward(self: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    K = self.in_features
    if (x.is_cuda and x.dtype == torch.bfloat16 and self.weight.dtype == torch.bfloat16 and not torch.is_grad_enabled()
            and (x.numel() // K <= MAX_ROWS or (x.numel() // K <= MAX_WIDE_ROWS and self.out_features >= WIDE_N)) and x.shape[-1] == K
            and (K % 64 == 0 or (x.numel() // K <= 4 and K % 8 == 0))):      # gemm_dispatch: K % 64 == 0 beyond the 4-row GEMV
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, N = x2.shape[0], self.out_features
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
        if (FMA_GEMV_ROWS < M <= MAX_ROWS16 or (MAX_ROWS16 < M <= MAX_STREAM_ROWS and N >= PACK64_MIN_N)) and K % 32 == 0:
            # 3..16 token rows (batched decode): the weight-streaming MFMA Linear, every wave streaming its own 16 weight rows;
            # 17..64 rows x a vocabulary-sized matrix (lm_head): ll_linear_rows64_bf16 on a packed copy (the decoder layers' own
            # Linears get theirs in _FusedLayer; any other module keeps the ring GEMM below rather than a second copy of its weight)
            bias = None
            if self.bias is not None:
                bias = getattr(self, "_ll_bias_f32", None)
                if bias is None or bias.device != x.device:
                    bias = self.bias.detach().float().contiguous()
                    self._ll_bias_f32, self._ll_bias_key = bias, _versions(self.bias)
            if M > MAX_ROWS16:
                wp = self.__dict__.get("_ll_w64")
                if wp is None:
                    wp = self._ll_w64 = _Packed64(self._ll_lib, self.weight.detach(), _versions(self.weight))
                return _rows64(self._ll_lib, x2, K, wp, bias, None, N, 0).reshape(*x.shape[:-1], N)
            rc = self._ll_lib.ll_linear_rows16_bf16(x2.data_ptr(), K, self.weight.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                                                    None, 0.0, None, 0, out.data_ptr(), N, M, N, K, 0, torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                _lib.check(rc, "ll_linear_rows16_bf16")
            return out.reshape(*x.shape[:-1], N)
        bias = None
        if self.bias is not None:
            bias = getattr(self, "_ll_bias_f32", None)
            if bias is None or bias.device != x.device:
                bias = self.bias.detach().float().contiguous()
                self._ll_bias_f32, self._ll_bias_key = bias, _versions(self.bias)
        splits = _prefill_splits(M, N, K)
        if splits > 1:
            ws = torch.empty(splits * M * N, dtype=torch.float32, device=x.device)
            rc = self._ll_lib.ll_linear_splitk_bf16(x2.data_ptr(), K, self.weight.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                                                    out.data_ptr(), N, M, N, K, 0, splits, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                _lib.check(rc, "ll_linear_splitk_bf16")
            return out.reshape(*x.shape[:-1], N)
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


def refresh_weight_copies(model: nn.Module) -> int:
    """The fused decode path keeps its own copies of some weights (q|k|v and gate|up concatenated once, biases in f32); a
    captured decode graph holds their ADDRESSES.  After the source weights changed (an SFT step, a merged LoRA adapter,
    load_state_dict) the copies are rewritten IN PLACE here -- same storage, so captured graphs stay valid -- keyed on
    (data_ptr, _version) of the sources, recorded when each copy is made.  GraphedDecoder.generate calls this before every
    generation (a few hundred tuple compares).  Returns the number of copies rewritten.  The copies cost about one extra set of
    MLP gate/up + attention q/k/v weights in memory (~8.5 GB for Qwen2-7B in bf16)."""
    n = 0
    with torch.no_grad():
        for mod in model.modules():
            d = mod.__dict__
            q = getattr(mod, "q_proj", None)
            grp = q.__dict__.get("_ll_qkv") if isinstance(q, nn.Module) else None
            if grp is not None and isinstance(getattr(mod, "k_proj", None), nn.Module):
                k, v = mod.k_proj, mod.v_proj
                key = _versions(q.weight, k.weight, v.weight, q.bias, k.bias, v.bias)
                if grp.key != key:
                    grp.w.copy_(torch.cat([q.weight.detach(), k.weight.detach(), v.weight.detach()], dim=0))
                    if grp.bias is not None:
                        grp.bias.copy_(torch.cat([(m.bias.detach().float() if m.bias is not None else
                                                   torch.zeros(m.out_features, device=grp.bias.device)) for m in (q, k, v)]))
                    grp.key = key
                    n += 1
            if "_ll_gate_up" in d and isinstance(getattr(mod, "gate_proj", None), nn.Module):
                key = _versions(mod.gate_proj.weight, mod.up_proj.weight)
                if d.get("_ll_gate_up_key") != key:
                    d["_ll_gate_up"].copy_(torch.cat([mod.gate_proj.weight.detach(), mod.up_proj.weight.detach()], dim=0))
                    d["_ll_gate_up_key"] = key
                    n += 1
            if "_ll_bias_f32" in d and getattr(mod, "bias", None) is not None:
                key = _versions(mod.bias)
                if d.get("_ll_bias_key") != key:
                    d["_ll_bias_f32"].copy_(mod.bias.detach().float())
                    d["_ll_bias_key"] = key
                    n += 1
            st = d.get("_ll_fused")
            if st is not None and st.bo is not None:
                key = _versions(mod.self_attn.o_proj.bias)
                if st.bo_key != key:
                    st.bo.copy_(mod.self_attn.o_proj.bias.detach().float())
                    st.bo_key = key
                    n += 1
            wp = d.get("_ll_w64")
            if wp is not None and getattr(mod, "weight", None) is not None:
                n += int(wp.sync(mod._ll_lib, mod.weight.detach(), _versions(mod.weight)))
        # the packed copies of the fused layers are made from the concatenated copies refreshed above
        for mod in model.modules():
            st = mod.__dict__.get("_ll_fused")
            if st is not None and st.p64 is not None:
                n += int(st.packed64(sync=True))
    return n


def drop_weight_copies(model: nn.Module) -> None:
    """Forget the concatenated / converted weight copies (restore_* paths): the next install rebuilds them."""
    for mod in model.modules():
        for k in ("_ll_gate_up", "_ll_gate_up_key", "_ll_bias_f32", "_ll_bias_key", "_ll_qkv", "_ll_w64"):
            mod.__dict__.pop(k, None)


def restore_linears(model: nn.Module) -> None:
    for mod in model.modules():
        if type(mod) is nn.Linear and "forward" in mod.__dict__:
            del mod.__dict__["forward"]


# ------------------------------------------------------------------------------------------ fused elementwise ops
def _decode_shaped(x: torch.Tensor, width: int, max_rows: int = MAX_ROWS) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and x.shape[-1] == width
            and x.numel() // width <= max_rows)


def _rmsnorm_forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
    H = self.weight.shape[0]
    if _decode_shaped(hidden_states, H, MAX_EW_ROWS) and self.weight.dtype == torch.bfloat16 and H % 8 == 0 and H <= 8192:
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


def _prefill_splits(M: int, N: int, K: int) -> int:
    """65..128 token rows x a matrix of at most ~6 k output rows (q|k|v, o_proj, down_proj of a 7-8 B model at prefill): 64 x 64 output tiles
    number fewer than half the CUs' worth of workgroups, so K is split in two f32 slabs + a slab sum (tools/prefill_splitk_sweep.py, 128 rows:
    down_proj 76.7 -> 51.7 us, o_proj 18.4 -> 15.1, q|k|v 20.3 -> 18.7 incl. the sum; gate|up with its 296+ tiles loses: one piece)."""
    if PREFILL_SPLITK and 64 < M <= MAX_ROWS and K % 128 == 0 and K >= 2048 and 2 * ((N + 63) // 64) < 200:
        return 2
    return 1


def _gemv(lib, x2: torch.Tensor, w: torch.Tensor, bias, N: int) -> torch.Tensor:
    M, K = x2.shape
    out = torch.empty(M, N, dtype=torch.bfloat16, device=x2.device)
    splits = _prefill_splits(M, N, K)
    if splits > 1:
        ws = torch.empty(splits * M * N, dtype=torch.float32, device=x2.device)
        rc = lib.ll_linear_splitk_bf16(x2.data_ptr(), K, w.data_ptr(), K, bias.data_ptr() if bias is not None else None, out.data_ptr(), N, M, N, K,
                                       0, splits, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_linear_splitk_bf16")
        return out
    rc = lib.ll_linear(_lib.LL_BF16, x2.data_ptr(), K, w.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                       out.data_ptr(), N, M, N, K, 0, 0, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        _lib.check(rc, "ll_linear")
    return out


def _mlp_forward(self, x: torch.Tensor) -> torch.Tensor:
    K, I = self.gate_proj.in_features, self.gate_proj.out_features
    if (_decode_shaped(x, K) and I % 8 == 0 and self.gate_proj.weight.dtype == torch.bfloat16 and self.gate_proj.bias is None
            and self.up_proj.bias is None):
        w = getattr(self, "_ll_gate_up", None)
        if w is None:   # gate and up share the input: one [2I, K] weight, one weight-streaming launch
            w = torch.cat([self.gate_proj.weight.detach(), self.up_proj.weight.detach()], dim=0).contiguous()
            self._ll_gate_up, self._ll_gate_up_key = w, _versions(self.gate_proj.weight, self.up_proj.weight)
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        gu = _gemv(self._ll_lib, x2, w, None, 2 * I)
        h = torch.empty(x2.shape[0], I, dtype=torch.bfloat16, device=x.device)
        rc = self._ll_lib.ll_silu_mul_bf16(gu.data_ptr(), gu.data_ptr() + 2 * I, h.data_ptr(), x2.shape[0], I, 2 * I,
                                           torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_silu_mul_bf16")
        return self.down_proj(h.reshape(*x.shape[:-1], I))
    if (_decode_shaped(x, K, MAX_EW_ROWS) and I % 8 == 0 and self.gate_proj.weight.dtype == torch.bfloat16):
        # prefill-sized call: BLAS projections, act_fn(gate) * up as one launch
        g, u = self.gate_proj(x), self.up_proj(x)
        g2, u2 = g.reshape(-1, I), u.reshape(-1, I)
        if g2.is_contiguous() and u2.is_contiguous():
            h = torch.empty_like(g2)
            rc = self._ll_lib.ll_silu_mul_bf16(g2.data_ptr(), u2.data_ptr(), h.data_ptr(), g2.shape[0], I, I,
                                               torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                _lib.check(rc, "ll_silu_mul_bf16")
            return self.down_proj(h.reshape(g.shape))
        return self.down_proj(self.act_fn(g) * u)
    return self._ll_orig_forward(x)


class _QKVGroup:
    """q_proj / k_proj / v_proj of one attention block share their input: the q_proj call runs ONE fused GEMV over the
    concatenated [q|k|v] weight and the k_proj / v_proj calls that follow on the same tensor return views of its result."""

    def __init__(self, lib, q, k, v):
        self.lib = lib
        self.nq, self.nk, self.nv = q.out_features, k.out_features, v.out_features
        self.K = q.in_features
        self.w = torch.cat([q.weight.detach(), k.weight.detach(), v.weight.detach()], dim=0).contiguous()
        biases = [m.bias for m in (q, k, v)]
        self.bias = None
        if any(b is not None for b in biases):
            self.bias = torch.cat([(b.detach().float() if b is not None else torch.zeros(m.out_features, device=q.weight.device))
                                   for b, m in zip(biases, (q, k, v))]).contiguous()
        self.x = None
        self.buf = None
        self.key = _versions(q.weight, k.weight, v.weight, q.bias, k.bias, v.bias)

    def run(self, x):
        x2 = x.reshape(-1, self.K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        self.buf = _gemv(self.lib, x2, self.w, self.bias, self.nq + self.nk + self.nv)
        self.x = x

    def part(self, x, lo, n):
        return self.buf[:, lo:lo + n].reshape(*x.shape[:-1], n) if x.dim() == 2 else self.buf.view(*x.shape[:-1], -1)[..., lo:lo + n]


def _q_forward(self, x):
    g = self._ll_qkv
    if _decode_shaped(x, g.K):
        g.run(x)
        return g.part(x, 0, g.nq)
    g.x = None
    return self._ll_orig_forward(x)


def _k_forward(self, x):
    g = self._ll_qkv
    if g.x is x and g.buf is not None:
        return g.part(x, g.nq, g.nk)
    return self._ll_orig_forward(x)


def _v_forward(self, x):
    g = self._ll_qkv
    if g.x is x and g.buf is not None:
        out = g.part(x, g.nq + g.nk, g.nv)
        g.x = None          # one use per attention call
        return out
    return self._ll_orig_forward(x)


def fuse_qkv(model: nn.Module) -> int:
    """Fuse the q/k/v projections of every attention block (modules with q_proj, k_proj, v_proj Linears) for decode-shaped
    calls.  Call after accelerate_linears (the non-decode path keeps whatever forward the Linears had)."""
    lib = _lib.load()
    n = 0
    for mod in model.modules():
        if all(type(getattr(mod, a, None)) is nn.Linear for a in ("q_proj", "k_proj", "v_proj")):
            q, k, v = mod.q_proj, mod.k_proj, mod.v_proj
            if q.weight.dtype != torch.bfloat16 or not q.weight.is_cuda or hasattr(q, "_ll_qkv"):
                continue
            grp = _QKVGroup(lib, q, k, v)
            for m, f in ((q, _q_forward), (k, _k_forward), (v, _v_forward)):
                m._ll_qkv = grp
                m._ll_orig_forward = m.forward
                m.forward = types.MethodType(f, m)
            n += 1
    return n


def _make_rope(orig, lib):
    import ctypes as C
    I64x3 = C.c_int64 * 3

    def apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=1, **kw):
        if (unsqueeze_dim == 1 and q.dim() == 4 and q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16
                and cos.dtype == torch.bfloat16 and cos.dim() == 3 and not torch.is_grad_enabled() and q.shape[2] <= MAX_EW_ROWS
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
    n_qkv = fuse_qkv(model)
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
    return {"rmsnorm": n_norm, "mlp": n_mlp, "rope": rope, "qkv": n_qkv}


def restore_elementwise(model: nn.Module) -> None:
    import sys
    for mod in model.modules():
        if "_ll_orig_forward" in mod.__dict__ and "forward" in mod.__dict__:
            del mod.__dict__["forward"]
            del mod.__dict__["_ll_orig_forward"]
    drop_weight_copies(model)            # a later install rebuilds them from the then-current weights
    base = getattr(model, "model", model)
    m = sys.modules.get(type(base).__module__)
    if m is not None and hasattr(getattr(m, "apply_rotary_pos_emb", None), "_ll_orig"):
        m.apply_rotary_pos_emb = m.apply_rotary_pos_emb._ll_orig


# ------------------------------------------------------------------------------------------ decode attention + KV append
ATTN_NAME = "llamole_decode"


def _decode_attention_forward(module, query, key, value, attention_mask, dropout=0.0, scaling=None, **kwargs):
    """AttentionInterface entry: softmax(q K^T * scaling + mask) V with grouped-query heads straight from the static cache
    (no repeat_kv copies); anything that is not a small bf16 decode call with a boolean mask goes to HF's sdpa path."""
    import ctypes as C
    from transformers.integrations.sdpa_attention import sdpa_attention_forward
    B, nh, S, D = query.shape
    if (query.is_cuda and query.dtype == torch.bfloat16 and key.dtype == torch.bfloat16 and S <= MAX_APPEND_ROWS and D in (64, 128)
            and attention_mask is not None and attention_mask.dtype == torch.bool and attention_mask.dim() == 4
            and attention_mask.shape[1] == 1 and attention_mask.shape[-1] == key.shape[2] and attention_mask.stride(3) == 1
            and key.is_contiguous() and value.is_contiguous() and query.stride(3) == 1 and dropout == 0.0
            and not torch.is_grad_enabled() and nh % key.shape[1] == 0):
        nkv, maxlen = key.shape[1], key.shape[2]
        out = torch.empty((B, S, nh, D), dtype=query.dtype, device=query.device)
        scale = float(scaling) if scaling is not None else D ** -0.5
        I64x3, I64x2 = C.c_int64 * 3, C.c_int64 * 2
        rc = _lib.load().ll_decode_attn_bf16(query.data_ptr(), key.data_ptr(), value.data_ptr(), attention_mask.data_ptr(),
                                             out.data_ptr(), B, nh, nkv, S, maxlen, D, scale,
                                             I64x3(query.stride(0), query.stride(1), query.stride(2)),
                                             I64x2(attention_mask.stride(0), attention_mask.stride(2)),
                                             torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_decode_attn_bf16")
        return out, None
    return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, **kwargs)


def use_decode_attention(model: nn.Module) -> bool:
    """Register the fused decode attention with transformers' AttentionInterface (boolean sdpa-style masks) and select it
    on ``model``.  Returns False (and changes nothing) if this transformers build has no such registry."""
    try:
        from transformers import AttentionInterface
        from transformers.masking_utils import AttentionMaskInterface, sdpa_mask
    except ImportError:
        return False
    AttentionInterface.register(ATTN_NAME, _decode_attention_forward)
    AttentionMaskInterface.register(ATTN_NAME, sdpa_mask)
    model.config._attn_implementation = ATTN_NAME
    for mod in model.modules():
        cfg = getattr(mod, "config", None)
        if cfg is not None and hasattr(cfg, "_attn_implementation"):
            cfg._attn_implementation = ATTN_NAME
    return True


def fuse_cache_update(cache) -> int:
    """Replace StaticLayer.update of an (already initialised) StaticCache by one fused append launch per layer for decode-
    shaped calls.  All layers write at the position held by layer 0's on-device counter; the caller advances that counter
    once per forward (GraphedDecoder does).  Returns the number of patched layers."""
    import ctypes as C
    lib = _lib.load()
    layers = list(cache.layers)
    if not layers or not all(getattr(l, "is_initialized", False) and hasattr(l, "cumulative_length") for l in layers):
        return 0
    pos = layers[0].cumulative_length
    I64x3 = C.c_int64 * 3

    def make(layer):
        orig = layer.update

        def update(key_states, value_states, *a, **k):
            S = key_states.shape[-2]
            if (S <= MAX_APPEND_ROWS and key_states.dtype == torch.bfloat16 and layer.keys.dtype == torch.bfloat16 and key_states.is_cuda
                    and key_states.stride(3) == 1 and value_states.stride(3) == 1 and not torch.is_grad_enabled()):
                B, nkv, _, D = key_states.shape
                rc = lib.ll_kv_append_bf16(layer.keys.data_ptr(), layer.values.data_ptr(), key_states.data_ptr(),
                                           value_states.data_ptr(), pos.data_ptr(), B, nkv, S, layer.keys.shape[2], D,
                                           I64x3(key_states.stride(0), key_states.stride(1), key_states.stride(2)),
                                           I64x3(value_states.stride(0), value_states.stride(1), value_states.stride(2)),
                                           torch.cuda.current_stream().cuda_stream)
                if rc != 0:
                    _lib.check(rc, "ll_kv_append_bf16")
                return layer.keys, layer.values
            return orig(key_states, value_states, *a, **k)

        layer.update = update
        layer._ll_fused_update = True

    for l in layers:
        if not getattr(l, "_ll_fused_update", False):
            make(l)
    return len(layers)


# ------------------------------------------------------------------------------------------ whole decoder layer = 5 launches
class _FusedLayer:
    """Decode-step state of one HF decoder layer (Qwen2 / Llama / Mistral layout): concatenated q|k|v and gate|up weights
    (shared with fuse_qkv / the MLP fusion when those ran first), f32 biases, norm weights.  ``run`` is
    Qwen2DecoderLayer.forward (transformers modeling_qwen2.py) for ONE new token per sequence:
      qkv  = gemv(rmsnorm(h), Wqkv) ; a = rope+append+attention(qkv) ; h = h + gemv(a, Wo)
      act  = silu(gate)*up of gemv(rmsnorm(h), Wgate|up) ; h = h + gemv(act, Wdown)
    five launches: batch <= 2 (<= 4 when the sizes are not multiples of 32) on the FMA GEMVs (ll_gemv_fused_bf16), batch 3..16 on
    the weight-streaming MFMA Linear (ll_linear_rows16_bf16) -- both with the RMSNorm as prologue and the residual / SiLU*mul
    epilogues."""

    def __init__(self, lib, layer):
        att, mlp = layer.self_attn, layer.mlp
        self.lib = lib
        self.layer_idx = att.layer_idx
        q, k, v, o = att.q_proj, att.k_proj, att.v_proj, att.o_proj
        grp = getattr(q, "_ll_qkv", None) or _QKVGroup(lib, q, k, v)
        self.wqkv, self.bqkv = grp.w, grp.bias
        self.nq, self.nkv_dim = q.out_features, k.out_features
        self.D = att.head_dim
        self.nh, self.nkv = self.nq // self.D, self.nkv_dim // self.D
        self.H = q.in_features
        self.scaling = float(att.scaling)
        self.wo = o.weight.detach()
        self.bo = o.bias.detach().float().contiguous() if o.bias is not None else None
        self.bo_key = _versions(o.bias)
        wgu = getattr(mlp, "_ll_gate_up", None)
        if wgu is None:
            wgu = torch.cat([mlp.gate_proj.weight.detach(), mlp.up_proj.weight.detach()], dim=0).contiguous()
            mlp._ll_gate_up, mlp._ll_gate_up_key = wgu, _versions(mlp.gate_proj.weight, mlp.up_proj.weight)
        self.wgu = wgu
        self.I = mlp.gate_proj.out_features
        self.wdown = mlp.down_proj.weight.detach()
        self.n1, self.n2 = layer.input_layernorm, layer.post_attention_layernorm
        self.eps1, self.eps2 = float(self.n1.variance_epsilon), float(self.n2.variance_epsilon)
        self.stream_ok = self.H % 32 == 0 and self.nq % 32 == 0 and self.I % 32 == 0 and self.n1.weight.dtype == torch.bfloat16
        self._mods = (q, k, v, o, mlp.gate_proj, mlp.up_proj, mlp.down_proj)
        self.p64 = None         # packed copies of the four matrices for 17..64 rows (made at the first such step)

    def _keys64(self):
        q, k, v, o, g, u, d = self._mods
        return (_versions(q.weight, k.weight, v.weight), _versions(o.weight), _versions(g.weight, u.weight), _versions(d.weight))

    def packed64(self, sync: bool = False):
        """[q|k|v], o_proj, [gate|up], down_proj in MFMA operand order (ll_rows64_pack_bf16) -- one more copy of the layer's weights,
        made the first time 17..64 sequences are decoded together; ``sync`` re-packs in place what changed at the source."""
        keys = self._keys64()
        srcs = (self.wqkv, self.wo, self.wgu, self.wdown)
        if self.p64 is None:
            self.p64 = tuple(_Packed64(self.lib, w, key) for w, key in zip(srcs, keys))
        elif sync:
            return sum(p.sync(self.lib, w, key) for p, w, key in zip(self.p64, srcs, keys))
        return 0

    def eligible(self, h, mask, cache, pe) -> bool:
        if not (h.is_cuda and h.dtype == torch.bfloat16 and h.dim() == 3 and h.shape[1] == 1
                and (h.shape[0] <= 4 or (h.shape[0] <= MAX_STREAM_ROWS and self.stream_ok))
                and not torch.is_grad_enabled() and cache is not None and pe is not None and mask is not None):
            return False
        layers = getattr(cache, "layers", None)
        if layers is None or self.layer_idx >= len(layers):
            return False
        cl = layers[self.layer_idx]
        if not (getattr(cl, "_ll_fused_update", False) and cl.keys.dtype == torch.bfloat16 and cl.keys.is_contiguous()
                and cl.values.is_contiguous()):
            return False
        cos = pe[0]
        return (mask.dtype == torch.bool and mask.dim() == 4 and mask.shape[1] == 1 and mask.shape[2] == 1
                and mask.shape[3] == cl.keys.shape[2] and mask.stride(3) == 1 and cos.dtype == torch.bfloat16
                and cos.dim() == 3 and cos.shape[1] == 1 and cos.shape[2] == self.D and cos.stride(2) == 1
                and pe[1].stride() == cos.stride() and h.is_contiguous())

    def _gemv(self, x, w, bias, norm_w, eps, res, N, K, epi):
        M = x.shape[0]
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
        if M > FMA_GEMV_ROWS and self.stream_ok:
            rc = self.lib.ll_linear_rows16_bf16(x.data_ptr(), x.stride(0), w.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                                                norm_w.data_ptr() if norm_w is not None else None, eps,
                                                res.data_ptr() if res is not None else None, res.stride(0) if res is not None else 0,
                                                out.data_ptr(), N, M, N, K, epi, torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                _lib.check(rc, "ll_linear_rows16_bf16")
            return out
        rc = self.lib.ll_gemv_fused_bf16(x.data_ptr(), x.stride(0), w.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                                         norm_w.data_ptr() if norm_w is not None else None, eps,
                                         res.data_ptr() if res is not None else None, res.stride(0) if res is not None else 0,
                                         out.data_ptr(), N, M, N, K, epi, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_gemv_fused_bf16")
        return out

    def _attn(self, qkv, nqkv, mask, cache, pe, B, device):
        cl = cache.layers[self.layer_idx]
        pos = cache.layers[0].cumulative_length
        cos, sin = pe
        att = torch.empty(B, self.nq, dtype=torch.bfloat16, device=device)
        rc = self.lib.ll_decode_attn_rope_bf16(qkv.data_ptr(), nqkv, cos.data_ptr(), sin.data_ptr(),
                                               0 if cos.shape[0] == 1 else cos.stride(0), cl.keys.data_ptr(), cl.values.data_ptr(),
                                               pos.data_ptr(), mask.data_ptr(), mask.stride(0), att.data_ptr(), B, self.nh, self.nkv,
                                               cl.keys.shape[2], self.D, self.scaling, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_decode_attn_rope_bf16")
        return att

    def run64(self, h, mask, cache, pe, pre=None, next_norm=None):
        """17..64 sequences: seven launches on the packed weights -- q|k|v (row scale in its epilogue), rope + append + attention, o_proj
        (K split into f32 slabs), slab sum + residual + the post-attention pre-norm, gate|up (row scale) + SiLU*mul, down_proj (K
        split), slab sum + residual + the NEXT layer's input pre-norm.  A "pre-norm" is the pair (bf16(h * w_norm), per-chunk sums of
        h^2): the RMSNorm is finished by the consuming Linear's epilogue.  ``pre`` = that pair for this layer's input when the previous
        layer already produced it; returns (h_out, pre_next or None)."""
        B, H = h.shape[0], self.H
        x = h.view(B, H)
        self.packed64()
        pqkv, po, pgu, pdown = self.p64
        xs, ssq = pre if pre is not None else _prenorm64(self.lib, x, self.n1.weight)
        nqkv = self.nq + 2 * self.nkv_dim
        qkv = _rows64(self.lib, xs, xs.stride(0), pqkv, self.bqkv, None, nqkv, 0, row_ssq=ssq, eps=self.eps1)
        att = self._attn(qkv, nqkv, mask, cache, pe, B, h.device)
        h1, xs2, ssq2 = _rows64(self.lib, att, att.stride(0), po, self.bo, x, H, 1, next_norm=self.n2.weight)
        act = _rows64(self.lib, xs2, xs2.stride(0), pgu, None, None, self.I, 2, row_ssq=ssq2, eps=self.eps2)
        if next_norm is None:
            return _rows64(self.lib, act, act.stride(0), pdown, None, h1, H, 1).view(B, 1, H), None
        h2, xs3, ssq3 = _rows64(self.lib, act, act.stride(0), pdown, None, h1, H, 1, next_norm=next_norm)
        return h2.view(B, 1, H), (xs3, ssq3)

    def run(self, h, mask, cache, pe):
        B, H = h.shape[0], self.H
        if B > MAX_ROWS16:
            return self.run64(h, mask, cache, pe)[0]
        x = h.view(B, H)
        cl = cache.layers[self.layer_idx]
        pos = cache.layers[0].cumulative_length
        cos, sin = pe
        nqkv = self.nq + 2 * self.nkv_dim
        qkv = self._gemv(x, self.wqkv, self.bqkv, self.n1.weight, self.eps1, None, nqkv, H, 0)
        att = torch.empty(B, self.nq, dtype=torch.bfloat16, device=h.device)
        rc = self.lib.ll_decode_attn_rope_bf16(qkv.data_ptr(), nqkv, cos.data_ptr(), sin.data_ptr(),
                                               0 if cos.shape[0] == 1 else cos.stride(0), cl.keys.data_ptr(), cl.values.data_ptr(),
                                               pos.data_ptr(), mask.data_ptr(), mask.stride(0), att.data_ptr(), B, self.nh, self.nkv,
                                               cl.keys.shape[2], self.D, self.scaling, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_decode_attn_rope_bf16")
        h1 = self._gemv(att, self.wo, self.bo, None, 0.0, x, H, self.nq, 1)
        act = self._gemv(h1, self.wgu, None, self.n2.weight, self.eps2, None, self.I, H, 2)
        h2 = self._gemv(act, self.wdown, None, None, 0.0, h1, H, self.I, 1)
        return h2.view(B, 1, H)


    def run_suffix(self, h, mask, cache, pe, B, S):
        """S consecutive new positions per sequence (B*S <= 16 rows: the query-token forward on the decode's cache): the five launches of
        ``run`` with ll_suffix_attn_rope_bf16 in the middle -- query row s sees the cache up to slot pos + s, the keys of rows 0..s included."""
        H, R = self.H, B * S
        x = h.view(R, H)
        cl = cache.layers[self.layer_idx]
        pos = cache.layers[0].cumulative_length
        cos, sin = pe
        nqkv = self.nq + 2 * self.nkv_dim
        qkv = self._gemv(x, self.wqkv, self.bqkv, self.n1.weight, self.eps1, None, nqkv, H, 0)
        att = torch.empty(R, self.nq, dtype=torch.bfloat16, device=h.device)
        rc = self.lib.ll_suffix_attn_rope_bf16(qkv.data_ptr(), nqkv, cos.data_ptr(), sin.data_ptr(), cl.keys.data_ptr(), cl.values.data_ptr(),
                                               pos.data_ptr(), mask.data_ptr(), att.data_ptr(), B, S, self.nh, self.nkv, cl.keys.shape[2],
                                               self.D, self.scaling, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_suffix_attn_rope_bf16")
        h1 = self._gemv(att, self.wo, self.bo, None, 0.0, x, H, self.nq, 1)
        act = self._gemv(h1, self.wgu, None, self.n2.weight, self.eps2, None, self.I, H, 2)
        h2 = self._gemv(act, self.wdown, None, None, 0.0, h1, H, self.I, 1)
        return h2.view(B, S, H)


def _layer_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                   position_embeddings=None, **kwargs):
    st = self._ll_fused
    if st.eligible(hidden_states, attention_mask, past_key_values, position_embeddings):
        return st.run(hidden_states, attention_mask, past_key_values, position_embeddings)
    return self._ll_layer_orig(hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                               past_key_values=past_key_values, use_cache=use_cache, position_embeddings=position_embeddings,
                               **kwargs)


def fuse_decoder_layers(model: nn.Module) -> int:
    """Run every decoder layer of a Qwen2 / Llama / Mistral-layout HF model as five launches at decode (batch <= 16), static cache with the fused append of ``fuse_cache_update``, boolean sdpa-style mask.  Any other call --
    prefill, larger batches, a dynamic cache -- takes the layer's previous forward.  Returns the number of patched layers."""
    lib = _lib.load()
    base = getattr(model, "model", model)
    layers = getattr(base, "layers", None)
    if layers is None:
        return 0
    n = 0
    for layer in layers:
        att, mlp = getattr(layer, "self_attn", None), getattr(layer, "mlp", None)
        if att is None or mlp is None or hasattr(layer, "_ll_fused"):
            continue
        ok = (all(type(getattr(att, a, None)) is nn.Linear for a in ("q_proj", "k_proj", "v_proj", "o_proj"))
              and all(type(getattr(mlp, a, None)) is nn.Linear for a in ("gate_proj", "up_proj", "down_proj"))
              and _is_silu(getattr(mlp, "act_fn", None)) and hasattr(layer, "input_layernorm")
              and hasattr(layer, "post_attention_layernorm") and hasattr(layer.input_layernorm, "variance_epsilon")
              and att.q_proj.weight.dtype == torch.bfloat16 and att.q_proj.weight.is_cuda
              and getattr(att, "head_dim", 0) in (64, 128) and att.q_proj.in_features % 8 == 0
              and att.q_proj.in_features <= 8192 and mlp.gate_proj.bias is None and mlp.up_proj.bias is None
              and mlp.down_proj.bias is None and getattr(att, "sliding_window", None) is None
              and not hasattr(att, "q_norm"))
        if not ok:
            continue
        layer._ll_fused = _FusedLayer(lib, layer)
        layer._ll_layer_orig = layer.forward
        layer.forward = types.MethodType(_layer_forward, layer)
        n += 1
    return n


def restore_decoder_layers(model: nn.Module) -> None:
    base = getattr(model, "model", model)
    for layer in getattr(base, "layers", []):
        if "_ll_fused" in layer.__dict__:
            del layer.__dict__["forward"]
            del layer.__dict__["_ll_fused"]
            del layer.__dict__["_ll_layer_orig"]
            for sub in layer.modules():       # the layer's concatenated gate|up copy goes with it (rebuilt on the next install)
                sub.__dict__.pop("_ll_gate_up", None)
                sub.__dict__.pop("_ll_gate_up_key", None)


# ------------------------------------------------------------------------------------------ whole decode step of the base model
def _model_forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                   use_cache=None, **kwargs):
    """Qwen2Model.forward / LlamaModel.forward for ONE new token per sequence on the fused layers: embedding gather, one
    prologue launch (rotary cos/sin + causal/padding key mask, instead of ~12 ATen launches of rotary_emb and
    create_causal_mask), the five-launch layers, final norm.  Anything else goes to the original forward."""
    st = self._ll_decode
    layers = self.layers[: self.config.num_hidden_layers]
    if (input_ids is not None and inputs_embeds is None and input_ids.dim() == 2 and input_ids.shape[1] == 1
            and input_ids.shape[0] <= MAX_STREAM_ROWS and input_ids.is_cuda and not torch.is_grad_enabled() and past_key_values is not None
            and attention_mask is not None and attention_mask.dim() == 2 and attention_mask.dtype == torch.long
            and attention_mask.stride(1) == 1 and position_ids is not None and position_ids.shape == input_ids.shape
            and position_ids.dtype == torch.long and not kwargs.get("output_hidden_states") and not kwargs.get("output_attentions")
            and getattr(past_key_values, "layers", None) and all(getattr(l, "_ll_fused_update", False) for l in past_key_values.layers)
            and attention_mask.shape[1] == past_key_values.layers[0].keys.shape[2]
            and past_key_values.layers[0].keys.dtype == torch.bfloat16):
        B, maxlen, D = input_ids.shape[0], attention_mask.shape[1], st["D"]
        dev = input_ids.device
        h = self.embed_tokens(input_ids)
        if h.dtype == torch.bfloat16 and h.is_contiguous():
            cos = torch.empty(B, 1, D, dtype=torch.bfloat16, device=dev)
            sin = torch.empty(B, 1, D, dtype=torch.bfloat16, device=dev)
            mask = torch.empty(B, 1, 1, maxlen, dtype=torch.bool, device=dev)
            inv_freq = self.rotary_emb.inv_freq
            pos = past_key_values.layers[0].cumulative_length
            posid = position_ids.contiguous()
            rc = st["lib"].ll_decode_prologue(posid.data_ptr(), inv_freq.data_ptr(), float(self.rotary_emb.attention_scaling),
                                              attention_mask.data_ptr(), attention_mask.stride(0), pos.data_ptr(), cos.data_ptr(),
                                              sin.data_ptr(), mask.data_ptr(), B, D, maxlen, torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                _lib.check(rc, "ll_decode_prologue")
            pe = (cos, sin)
            if layers[0]._ll_fused.eligible(h, mask, past_key_values, pe):
                if B > MAX_ROWS16:
                    # 17..64 sequences: every layer's closing launch also produces the next layer's input pre-norm
                    pre = None
                    for i, layer in enumerate(layers):
                        nxt = layers[i + 1]._ll_fused if i + 1 < len(layers) else None
                        h, pre = layer._ll_fused.run64(h, mask, past_key_values, pe, pre, nxt.n1.weight if nxt is not None else None)
                else:
                    for layer in layers:
                        h = layer._ll_fused.run(h, mask, past_key_values, pe)
                h = self.norm(h)
                from transformers.modeling_outputs import BaseModelOutputWithPast
                return BaseModelOutputWithPast(last_hidden_state=h, past_key_values=past_key_values if use_cache else None)
    if st.get("suffix") and input_ids is not None and inputs_embeds is None and input_ids.dim() == 2:
        out = _suffix_forward(self, st, layers, input_ids, attention_mask, position_ids, past_key_values, use_cache, kwargs)
        if out is not None:
            return out
    return self._ll_model_orig(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                               past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache, **kwargs)


def _suffix_forward(self, st, layers, input_ids, attention_mask, position_ids, cache, use_cache, kwargs):
    """[B,S] new tokens (3 <= B*S <= 16) appended at the cache position held by layer 0's counter -- GraphedDecoder.continue_hidden, the
    reference's query-token re-forward -- on the five-launch layers: one prologue launch (cos/sin per row, key mask per row: slot
    j <= pos + s and not padding), then per layer q|k|v with the RMSNorm prologue, rope + append + attention, o_proj + residual, gate|up
    + SiLU*mul, down_proj + residual (ll_linear_rows16_bf16).  None = not this shape: the caller takes the HF forward."""
    B, S = input_ids.shape
    R = B * S
    clayers = getattr(cache, "layers", None)
    if not (S >= 2 and FMA_GEMV_ROWS < R <= MAX_ROWS16 and input_ids.is_cuda and not torch.is_grad_enabled() and clayers
            and all(getattr(l, "_ll_fused_update", False) for l in clayers) and attention_mask is not None and attention_mask.dim() == 2
            and attention_mask.dtype == torch.long and attention_mask.stride(1) == 1 and attention_mask.shape[0] == B
            and attention_mask.shape[1] == clayers[0].keys.shape[2] and clayers[0].keys.dtype == torch.bfloat16
            and clayers[0].keys.is_contiguous() and clayers[0].values.is_contiguous() and clayers[0].keys.shape[0] == B
            and position_ids is not None and position_ids.shape == input_ids.shape and position_ids.dtype == torch.long
            and not kwargs.get("output_hidden_states") and not kwargs.get("output_attentions")
            and all(l._ll_fused.stream_ok for l in layers)):
        return None
    h = self.embed_tokens(input_ids)
    if h.dtype != torch.bfloat16 or not h.is_contiguous():
        return None
    maxlen, D, dev = attention_mask.shape[1], st["D"], input_ids.device
    cos = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
    sin = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
    mask = torch.empty(R, maxlen, dtype=torch.bool, device=dev)
    posid = position_ids.contiguous()
    rc = st["lib"].ll_suffix_prologue(posid.data_ptr(), self.rotary_emb.inv_freq.data_ptr(), float(self.rotary_emb.attention_scaling),
                                      attention_mask.data_ptr(), attention_mask.stride(0), clayers[0].cumulative_length.data_ptr(),
                                      cos.data_ptr(), sin.data_ptr(), mask.data_ptr(), B, S, D, maxlen, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        _lib.check(rc, "ll_suffix_prologue")
    for layer in layers:
        h = layer._ll_fused.run_suffix(h, mask, cache, (cos, sin), B, S)
    h = self.norm(h)
    from transformers.modeling_outputs import BaseModelOutputWithPast
    return BaseModelOutputWithPast(last_hidden_state=h, past_key_values=cache if use_cache else None)


def suffix_on_fused_layers(model: nn.Module, on: bool) -> bool:
    """Switch for the query-token forward of ``fuse_model_decode``-patched models: while on, a [B,S] call with B*S <= 16 rows that
    continues a fused static cache runs on the five-launch layers (GraphedDecoder.continue_hidden turns it on around its call).
    Returns whether the model has the patch."""
    base = getattr(model, "model", model)
    st = base.__dict__.get("_ll_decode")
    if st is None:
        return False
    st["suffix"] = bool(on)
    return True


def fuse_model_decode(model: nn.Module) -> bool:
    """Run the base model's decode step without HF's per-token mask / rotary-table construction (call after
    ``fuse_decoder_layers``; needs every layer fused, default rope, f32 ``inv_freq``).  Returns whether it was installed."""
    base = getattr(model, "model", model)
    layers = getattr(base, "layers", None)
    rot = getattr(base, "rotary_emb", None)
    if (layers is None or rot is None or hasattr(base, "_ll_decode") or not all(hasattr(l, "_ll_fused") for l in layers)
            or not hasattr(rot, "inv_freq") or rot.inv_freq.dtype != torch.float32 or not rot.inv_freq.is_cuda
            or getattr(rot, "rope_type", "default") != "default" or not hasattr(base, "embed_tokens") or not hasattr(base, "norm")
            or getattr(base, "has_sliding_layers", False)):
        return False
    D = layers[0]._ll_fused.D
    if rot.inv_freq.numel() * 2 != D:
        return False
    base._ll_decode = {"lib": _lib.load(), "D": D}
    base._ll_model_orig = base.forward
    base.forward = types.MethodType(_model_forward, base)
    return True


def restore_model_decode(model: nn.Module) -> None:
    base = getattr(model, "model", model)
    if "_ll_decode" in base.__dict__:
        del base.__dict__["forward"]
        del base.__dict__["_ll_decode"]
        del base.__dict__["_ll_model_orig"]


# ------------------------------------------------------------------------------------------ one call for the whole stack
def accelerate_llm(model: nn.Module, linears: bool = True, fuse: bool = True, layers: bool = True, model_decode: bool = True) -> dict:
    """Apply the decode acceleration stack to an (untouched) HF Llama-family causal LM on a HIP device and report what took
    effect: GEMV under nn.Linear -> fused RMSNorm / rotary / SiLU*mul and q|k|v, gate|up fusion -> decode attention through
    the AttentionInterface registry -> five-launch decoder layers -> one-launch decode prologue.  Every stage falls back to
    the HF code for shapes or architectures it does not cover; pair it with ``GraphedDecoder(..., fused_cache=True)``."""
    info: dict = {"linears": 0}
    if not next(model.parameters()).is_cuda:
        return info
    if linears:
        info["linears"] = accelerate_linears(model)
    if linears and fuse:
        info.update(accelerate_elementwise(model))
        info["decode_attention"] = bool(use_decode_attention(model))
        if info["decode_attention"] and layers:
            info["decoder_layers_5_launches"] = fuse_decoder_layers(model)
            if model_decode:
                info["decode_prologue_1_launch"] = fuse_model_decode(model)
    return info
