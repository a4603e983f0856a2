"""HIP-graph autoregressive decode around an untouched HuggingFace causal LM.

Row f2 of SURVEY.md section 8 ("LLM-side hot loop"): the reference calls ``language_model.generate`` eagerly
(modeling_llamole.py:599, :849), i.e. ~10^3 small launches per token from Python.  On MI355X the decode step
of a 7-8 B model is launch-bound long before it is HBM-bound, so the step -- the stock HF ``forward`` over a
``StaticCache`` -- is captured ONCE as a hipGraph (``torch.cuda.CUDAGraph``; no tracing compiler, no Triton)
and replayed per token; sampling (temperature, top-p, multinomial) stays on the device -- on a HIP device with bf16
logits it is ONE launch (``ll_sample_token_bf16``) captured in the same graph together with the loop bookkeeping --
and the only host sync is an EOS check every ``sync_every`` tokens.  The LLM forward itself is HF code on
PyTorch-ROCm; ``llm_accel`` swaps HIP kernels in underneath its modules.

``GraphedDecoder.generate(input_ids, attention_mask, ...)`` returns prompt + new tokens like ``generate``
(rows that stopped are padded with ``pad_token_id``).  Greedy mode is token-identical to HF ``generate``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import os

import torch


def sample_top_p(logits: torch.Tensor, temperature: float, top_p: float, generator=None, top_k: Optional[int] = None) -> torch.Tensor:
    """HF TemperatureLogitsWarper + TopKLogitsWarper + TopPLogitsWarper + multinomial, without host syncs.  logits [B,V] float."""
    logits = logits.float()
    logits = torch.nan_to_num(logits, nan=0.0, posinf=torch.finfo(torch.float32).max, neginf=torch.finfo(torch.float32).min)
    if temperature is not None and temperature != 1.0:
        logits = logits / temperature
    if top_k is not None and 0 < top_k < logits.shape[-1]:
        kth = torch.topk(logits, top_k, dim=-1).values[..., -1:]
        logits = logits.masked_fill(logits < kth, float("-inf"))      # ties with the k-th value stay, as in HF
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_idx = torch.sort(logits, descending=False)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[..., -1:] = False                      # keep at least one token
        remove = remove.scatter(1, sorted_idx, remove)
        logits = logits.masked_fill(remove, float("-inf"))
    probs = torch.softmax(logits, dim=-1)
    return torch.multinomial(probs, 1, generator=generator).squeeze(1)


def _reject_unsupported_generation_options(other: dict) -> None:
    """The decoder implements temperature, top_k, top_p, greedy, eos / pad handling.  Anything else HF ``generate`` would honour
    must be at its neutral value -- a silently dropped option would sample from another distribution than the reference."""
    bad = []
    for k, v in other.items():
        if v is None:
            continue
        if k == "repetition_penalty" and float(v) == 1.0:
            continue
        if k == "length_penalty" and float(v) == 1.0:
            continue
        if k == "num_beams" and int(v) == 1:
            continue
        if k == "logits_processor":
            # the reference passes get_logits_processor() = [InfNanRemoveLogitsProcessor] (extras/misc.py:146-152): the samplers
            # here sanitise inf / nan themselves; any other processor changes the distribution
            if all(type(p).__name__ == "InfNanRemoveLogitsProcessor" for p in v):
                continue
        if k in ("use_cache", "return_dict_in_generate", "output_scores", "output_hidden_states", "synced_gpus"):
            if not v or k == "use_cache":
                continue
        bad.append(k)
    if bad:
        raise NotImplementedError(f"GraphedDecoder.generate does not implement {sorted(bad)}: set them to their neutral values or "
                                  f"use the HF path (llm_decode='hf')")


MAX_HIP_VOCAB = 163840   # ll_sample_token_bf16 keeps a whole row of logits in one workgroup's registers
N_EOS_SLOTS = 32


class GraphedDecoder:
    def __init__(self, model, use_graph: bool = True, sync_every: int = 16, fused_cache: bool = False, sampler: str = "hip"):
        self.model = model
        self.use_graph = use_graph and next(model.parameters()).is_cuda
        self.sync_every = sync_every
        self.fused_cache = fused_cache    # one fused KV-append launch per layer (llm_accel.fuse_cache_update)
        # "hip": on a HIP device with bf16 logits the sampler (temperature / top-p / multinomial, or argmax) and the loop
        # bookkeeping are ONE launch inside the captured step (ll_sample_token_bf16); "torch": op-by-op PyTorch sampler
        self.sampler = sampler
        self.len_bucket = 64
        # the eager query-token forward over the decode's KV cache (continue_hidden: <= 16 new positions, so the attention is this
        # library's decode kernel and HF's mask is the explicit one either way) is host-bound -- 28 layers x ~20 Python-dispatched ops,
        # 7.4 ms of host time for 3.7 ms of kernels at Qwen2-7B: a (batch, length) shape seen a second time is captured as a hipGraph
        # and replayed from then on (LLAMOLE_GRAPH_SUFFIX=0: always eager).  The prompt prefill is NOT captured: transformers takes
        # other mask decisions while a stream is capturing (masking_utils.is_tracing), i.e. other SDPA kernels and roundings than the
        # eager prefill, and under the overlapped trajectory the prefill is device-bound anyway (HISTORY R5.8).
        self.graph_suffix = self.use_graph and os.environ.get("LLAMOLE_GRAPH_SUFFIX", "1") != "0"
        self.split_sampler = os.environ.get("LLAMOLE_SPLIT_SAMPLER", "1") != "0"     # top-k sampling as two launches (candidates, finish)
        self.sample_ws = None
        self.max_side_graphs = 4
        self._side_graphs = {}
        self._sample_key = None
        self._cache_fused = False
        self._key = None
        self._graph = None

    # static buffers + captured step for a (batch, max_len) shape
    def _prepare(self, B: int, max_len: int, device, use_embeds: bool):
        from transformers import StaticCache
        key = (B, max_len, use_embeds)
        if self._key == key:
            self.cache.reset()
            return
        self._key = key
        cfg = self.model.config
        self.cache = StaticCache(config=cfg, max_cache_len=max_len)
        self.tok = torch.zeros(B, 1, dtype=torch.long, device=device)
        self.pos = torch.zeros(1, dtype=torch.long, device=device)
        self.mask = torch.zeros(B, max_len, dtype=torch.long, device=device)
        self.posid = torch.zeros(B, 1, dtype=torch.long, device=device)   # static: the captured graph reads it
        self.out_buf = torch.zeros(B, max_len, dtype=torch.long, device=device)
        self.done = torch.zeros(B, dtype=torch.uint8, device=device)
        self.stepc = torch.zeros(B, dtype=torch.long, device=device)
        self.seed_buf = torch.zeros(1, dtype=torch.long, device=device)
        self.eos_buf = torch.full((N_EOS_SLOTS,), -1, dtype=torch.long, device=device)
        self.logits = None
        self.sample_ws = None
        self._graph = None
        self._cache_fused = False
        self._side_graphs = {}          # captured against the buffers above

    def _captured(self, key, statics, fn):
        """``fn(*static buffers)`` eager the first time ``key`` is seen, captured the second time, replayed afterwards; ``statics`` are
        (static buffer factory, current value) pairs.  Returns fn's output (a tensor that the next replay overwrites).  A capture that
        fails (an op that synchronises) switches the shape back to eager for good."""
        values = [v for _, v in statics]
        st = self._side_graphs.get(key)
        if not self.graph_suffix or st == "eager":
            return fn(*values)
        if st is None:
            if len(self._side_graphs) >= self.max_side_graphs:      # shapes that do not repeat (A* expansion prompts) stay eager
                # evict a shape that was only SEEN, oldest first; captured graphs and "eager" tombstones stay (dropping a live graph means
                # a re-capture -- a device-wide synchronise inside the pipelined step -- and a forgotten tombstone retries a failed capture)
                victim = next((k for k, v in self._side_graphs.items() if v == "seen"), None)
                if victim is None:
                    return fn(*values)                              # table full of live graphs / tombstones: this shape runs eager
                self._side_graphs.pop(victim)
            self._side_graphs[key] = "seen"
            return fn(*values)
        if st == "seen":
            bufs = [mk(v) for (mk, _), v in zip(statics, values)]
            for b, v in zip(bufs, values):
                b.copy_(v)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = fn(*bufs)
            except Exception as e:          # noqa: BLE001 -- whatever the capture refuses: this shape runs eager from now on
                import warnings
                warnings.warn(f"hipGraph capture of the {key[0]} forward failed ({type(e).__name__}: {e}); it stays eager")
                self._side_graphs[key] = "eager"
                torch.cuda.synchronize()
                return fn(*values)
            st = self._side_graphs[key] = (g, bufs, out)
        g, bufs, out = st
        for b, v in zip(bufs, values):
            b.copy_(v)
        g.replay()
        return out

    def _step(self):
        out = self.model(input_ids=self.tok, attention_mask=self.mask, past_key_values=self.cache,
                         cache_position=self.pos, position_ids=self.posid, use_cache=True, return_dict=True)
        if self._cache_fused:   # the fused per-layer appends all used layer 0's counter; advance it once per forward
            self.cache.layers[0].cumulative_length.add_(self.tok.shape[1])
        return out.logits[:, -1, :]

    def _rewind(self, n: int):
        layers = self.cache.layers[:1] if self._cache_fused else self.cache.layers
        for layer in layers:
            layer.cumulative_length.sub_(n)

    @torch.no_grad()
    def continue_hidden(self, tail_ids: torch.Tensor, start: int) -> torch.Tensor:
        """Final hidden states of ``tail_ids`` [B,S] run ON TOP of the KV cache of the last ``generate`` call, at cache
        slots ``start .. start+S-1`` (slots >= start are overwritten).  This is the reference's query-token re-forward
        (modeling_llamole.py:641-646) without recomputing the prompt + analysis prefix."""
        info = self._last
        if info is None:
            raise RuntimeError("continue_hidden needs a preceding generate() on this decoder")
        B, S = tail_ids.shape
        P = info["P"]
        if start < P or start + S > self.mask.shape[1]:
            raise ValueError(f"continuation [{start},{start + S}) outside the cache window [{P},{self.mask.shape[1]})")
        for layer in self.cache.layers:
            layer.cumulative_length.fill_(start)
        device = tail_ids.device
        pos = torch.arange(start, start + S, device=device)
        posid = info["plen"] + (start - P) + torch.arange(S, device=device).unsqueeze(0)
        base = getattr(self.model, "model", self.model)

        def fwd(ids, cache_pos, pos_ids):
            return base(input_ids=ids, attention_mask=self.mask, past_key_values=self.cache, cache_position=cache_pos,
                        position_ids=pos_ids, use_cache=True, return_dict=True).last_hidden_state

        from .llm_accel import suffix_on_fused_layers
        fused = self._cache_fused and os.environ.get("LLAMOLE_FUSED_SUFFIX", "1") != "0" and suffix_on_fused_layers(self.model, True)
        try:
            hidden = self._captured(("suffix", B, S), [(torch.empty_like, tail_ids), (torch.empty_like, pos), (torch.empty_like, posid)], fwd)
        finally:
            if fused:
                suffix_on_fused_layers(self.model, False)
        if self._cache_fused:
            self.cache.layers[0].cumulative_length.add_(S)
        return hidden

    def _hip_sample(self, logits: torch.Tensor, sp, advance: int):
        """One launch: sample (or argmax) from bf16 logits [B,V], write tok / out_buf[:, step], update done / step and,
        with ``advance``, the position counters the next forward reads."""
        from . import _lib
        greedy, inv_temp, top_p, pad, top_k = sp
        B, V = logits.shape
        lib = _lib.load()
        if self.sample_ws is None or self.sample_ws.device != logits.device:
            # workspace of the split top-k sampler (candidate lists; zero-filled once, the sampler leaves it clean): a static buffer of the graph
            self.sample_ws = torch.zeros(int(lib.ll_sample_workspace_bytes(B)), dtype=torch.uint8, device=logits.device)
        rc = lib.ll_sample_token_topk_ws_bf16(logits.data_ptr(), logits.stride(0), B, V, inv_temp, top_p, int(top_k), int(greedy),
                                              self.seed_buf.data_ptr(), self.eos_buf.data_ptr(), N_EOS_SLOTS, pad,
                                              self.done.data_ptr(), self.tok.data_ptr(), self.out_buf.data_ptr(),
                                              self.out_buf.stride(0), self.out_buf.shape[1], self.stepc.data_ptr(),
                                              self.posid.data_ptr(), self.pos.data_ptr(), advance, None,
                                              self.sample_ws.data_ptr() if self.split_sampler else None, self.sample_ws.numel(),
                                              torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            _lib.check(rc, "ll_sample_token_topk_ws_bf16")

    def _generate_hip(self, logits, sp, P, plen, eos_list, max_new_tokens, generator, device):
        """Decode loop with the fused sampler: per token the host only replays ONE graph (forward + sampler)."""
        self.done.zero_()
        self.stepc.zero_()
        self.out_buf.fill_(sp[3])
        self.eos_buf.fill_(-1)
        if eos_list:
            self.eos_buf[:len(eos_list)] = torch.tensor(eos_list, dtype=torch.long, device=device)
        gen_dev = generator.device.type if generator is not None else device.type
        self.seed_buf.copy_(torch.randint(0, 2 ** 62, (1,), device=gen_dev, generator=generator))
        self.posid.copy_(plen)
        self.pos.fill_(P)
        if self._sample_key != sp:          # sampler parameters are baked into the captured launch
            self._sample_key = sp
            self._graph = None
        self._hip_sample(logits, sp, 0)
        from ._trace import mark
        mark("generate: first token sampled")
        n = 1
        for t in range(1, max_new_tokens):
            if eos_list and t % self.sync_every == 0:
                if bool(self.done.all()):
                    break
                mark("generate: rendezvous")
            if self.use_graph:
                if self._graph is None:
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        self._step()                              # warm-up of the forward only; state rewound below
                        self._rewind(self.tok.shape[1])
                    torch.cuda.current_stream().wait_stream(s)
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                        self.logits = self._step()
                        self._hip_sample(self.logits, sp, 1)
                self._graph.replay()
                logits = self.logits
            else:
                logits = self._step()
                self._hip_sample(logits, sp, 1)
            n = t + 1
        self.last_logits = logits
        self._last["n_new"] = n
        return self.out_buf[:, :n].clone()

    @torch.no_grad()
    def generate(self, input_ids: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                 inputs_embeds: Optional[torch.Tensor] = None, max_new_tokens: int = 128, do_sample: bool = True,
                 temperature: float = 1.0, top_p: float = 1.0, eos_token_id: Optional[Sequence[int]] = None,
                 pad_token_id: Optional[int] = None, generator=None, top_k: Optional[int] = None, **other) -> torch.Tensor:
        _reject_unsupported_generation_options(other)
        from ._trace import mark
        mark("generate: enter")
        top_k = int(top_k) if top_k else 0
        from .llm_accel import refresh_weight_copies
        refresh_weight_copies(self.model)       # concatenated / converted weight copies follow their sources (in place)
        ref = input_ids if input_ids is not None else inputs_embeds
        B, P = ref.shape[0], ref.shape[1]
        device = ref.device
        if attention_mask is None:
            attention_mask = torch.ones(B, P, dtype=torch.long, device=device)
        # static-cache length rounded up to a bucket: prompts of slightly different lengths (A* expansion prompts, eval
        # batches) then share one cache allocation and ONE captured graph; unused tail slots stay masked by causality
        max_len = -(-(P + max_new_tokens) // self.len_bucket) * self.len_bucket
        self._prepare(B, max_len, device, inputs_embeds is not None)
        self._last = None
        eos = torch.tensor(list(eos_token_id) if isinstance(eos_token_id, (list, tuple)) else
                           ([] if eos_token_id is None else [eos_token_id]), dtype=torch.long, device=device)
        pad = pad_token_id if pad_token_id is not None else (int(eos[0]) if eos.numel() else 0)
        # ---- prefill (eager, one call)
        # 2-D key mask for the whole static cache: prompt padding as given, every future slot open (causality
        # -- cache_position inside the HF mask builder -- already hides the slots not written yet), so the mask
        # never changes between replays of the captured step.
        self.mask.fill_(1)
        self.mask[:, :P] = attention_mask
        plen = attention_mask.long().sum(dim=1, keepdim=True)            # valid prompt tokens per row (left padding)
        pos_ids = (attention_mask.long().cumsum(dim=1) - 1).clamp_min(0)
        kw = dict(inputs_embeds=inputs_embeds) if inputs_embeds is not None else dict(input_ids=input_ids)
        pre = dict(attention_mask=self.mask[:, :P], past_key_values=self.cache, cache_position=torch.arange(P, device=device),
                   position_ids=pos_ids, use_cache=True, return_dict=True, **kw)
        try:
            out = self.model(logits_to_keep=1, **pre)       # only the last position's logits are used
        except TypeError:                                   # a model class without that argument
            out = self.model(**pre)
        logits = out.logits[:, -1, :]
        mark("generate: prefill enqueued")
        for layer in self.cache.layers:          # the next free slot is P whichever update path the prefill took (a short
            if hasattr(layer, "cumulative_length"):   # prompt on a re-used cache goes through the fused append, which
                layer.cumulative_length.fill_(P)      # leaves advancing layer 0's shared counter to its caller)
        self._last = dict(P=P, plen=plen, max_new=max_new_tokens, from_ids=input_ids is not None)
        if self.fused_cache and not self._cache_fused and device.type == "cuda":
            from .llm_accel import fuse_cache_update
            self._cache_fused = fuse_cache_update(self.cache) > 0
        if (self.sampler == "hip" and logits.is_cuda and logits.dtype == torch.bfloat16 and logits.shape[1] % 8 == 0
                and logits.shape[1] <= MAX_HIP_VOCAB and logits.stride(1) == 1 and eos.numel() <= N_EOS_SLOTS
                and (not do_sample or (temperature or 1.0) > 0)):
            import numpy as np
            temp = 1.0 if temperature is None else float(temperature)
            sp = (not do_sample, float(np.float32(1.0) / np.float32(temp)), 1.0 if top_p is None else float(top_p), int(pad), top_k)
            new_tokens = self._generate_hip(logits, sp, P, plen, eos.tolist(), max_new_tokens, generator, device)
            mark("generate: decode loop done")
            return torch.cat([input_ids, new_tokens], dim=1) if input_ids is not None else new_tokens
        if self._sample_key is not None:     # a graph captured with the fused sampler does not fit the torch-sampler loop
            self._sample_key = None
            self._graph = None
        new_tokens = torch.full((B, max_new_tokens), pad, dtype=torch.long, device=device)
        done = torch.zeros(B, dtype=torch.bool, device=device)
        self.posid.copy_(plen)                                           # position id of the next token, per row
        n_done_steps = 0
        for t in range(max_new_tokens):
            nxt = sample_top_p(logits, temperature, top_p, generator, top_k) if do_sample else logits.argmax(dim=-1)
            nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
            new_tokens[:, t] = nxt
            if eos.numel():
                done = done | torch.isin(nxt, eos)
            n_done_steps = t + 1
            if t + 1 == max_new_tokens:
                break
            if eos.numel() and (t + 1) % self.sync_every == 0 and bool(done.all()):
                break
            # ---- one decode step at cache position P + t
            self.tok.copy_(nxt.view(B, 1))
            self.pos.fill_(P + t)
            if t > 0:
                self.posid.add_(1)
            if self.use_graph:
                if self._graph is None:
                    # warm-up on a side stream, then capture the stock HF forward once
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        self._step()
                        # the warm-up advanced the cache's on-device length counters; rewind them so that the
                        # captured step (replayed below for this same token) writes the same slot again
                        self._rewind(self.tok.shape[1])
                    torch.cuda.current_stream().wait_stream(s)
                    self._graph = torch.cuda.CUDAGraph()
                    # thread-local capture: a RCCL watchdog thread polling events must not invalidate the capture
                    with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                        self.logits = self._step()
                self._graph.replay()
                logits = self.logits
            else:
                logits = self._step()
        new_tokens = new_tokens[:, :n_done_steps]
        self._last["n_new"] = n_done_steps
        self.last_logits = logits
        if input_ids is not None:
            return torch.cat([input_ids, new_tokens], dim=1)
        return new_tokens
