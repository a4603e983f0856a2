"""End-to-end synthetic workload: BASELINE.json configs[1] (Qwen2-7B + GraphDiT material design).

There is no network on the GPU box, so the LLM is the *architecture* named by the config with seeded
random-init weights (HuggingFace ``Qwen2ForCausalLM`` on PyTorch-ROCm, bf16, untouched), the prompt is a
synthetic ``cutoff_len``-token id tensor, and generation uses the reference's sampling settings
(config/generate/qwen_material.yaml: temperature 0.6, top_p 0.9, max_new_tokens 128; stop at eos or any of
the 9 special tokens, eval/workflow.py:93-98).  With random weights the trigger token is (almost) never
sampled, so every prompt pays the full max_new_tokens budget -- the worst case of the real workload.
"""
from __future__ import annotations

import time
import types

import torch

LLM_CONFIGS = {
    # published architecture hyper-parameters of the base models named in BASELINE.json
    "qwen2-7b": dict(cls="Qwen2", hidden_size=3584, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                     intermediate_size=18944, vocab_size=152064, max_position_embeddings=32768, rope_theta=1000000.0,
                     rms_norm_eps=1e-6, tie_word_embeddings=False),
    "llama-3.1-8b": dict(cls="Llama", hidden_size=4096, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
                         intermediate_size=14336, vocab_size=128256, max_position_embeddings=131072, rope_theta=500000.0,
                         rms_norm_eps=1e-5, tie_word_embeddings=False),
    "mistral-7b": dict(cls="Mistral", hidden_size=4096, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
                       intermediate_size=14336, vocab_size=32768, max_position_embeddings=32768, rope_theta=1000000.0,
                       rms_norm_eps=1e-5, sliding_window=None, tie_word_embeddings=False),
    "tiny-llama": dict(cls="Llama", hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=512, vocab_size=2048, max_position_embeddings=2048, rope_theta=10000.0,
                       rms_norm_eps=1e-5, tie_word_embeddings=False),
    "tiny-mistral": dict(cls="Mistral", hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                         intermediate_size=512, vocab_size=2048, max_position_embeddings=2048, rope_theta=10000.0,
                         rms_norm_eps=1e-5, sliding_window=None, tie_word_embeddings=False),
    "tiny": dict(cls="Qwen2", hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                 intermediate_size=512, vocab_size=2048, max_position_embeddings=2048, rope_theta=10000.0,
                 rms_norm_eps=1e-6, tie_word_embeddings=False),
}


class SyntheticTokenizer:
    """Just enough tokenizer for synthetic-id workloads: eos id, special-token ids, trivial decode."""

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size
        self.eos_token_id = vocab_size - 20
        self.pad_token_id = self.eos_token_id
        from .modeling_llamole import SPECIAL_TOKENS
        self.special = {t: vocab_size - 19 + i for i, t in enumerate(SPECIAL_TOKENS)}

    def __len__(self):
        return self.vocab_size

    def encode(self, text, add_special_tokens=False, return_tensors=None):
        """3 characters per token (stable across processes); special-token strings inside the text become their ids."""
        import re
        import zlib
        if getattr(self, "_split", None) is None:
            self._split = re.compile("(" + "|".join(re.escape(t) for t in sorted(self.special, key=len, reverse=True)) + ")")
        ids = []
        for piece in self._split.split(text):
            if not piece:
                continue
            if piece in self.special:
                ids.append(self.special[piece])
                continue
            data = piece.encode()
            if len(data) == len(piece):       # ASCII: chunk the bytes directly
                ids.extend(5 + zlib.crc32(data[i:i + 3]) % 1000 for i in range(0, len(data), 3))
            else:
                ids.extend(5 + zlib.crc32(piece[i:i + 3].encode()) % 1000 for i in range(0, len(piece), 3))
        return torch.tensor([ids]) if return_tensors == "pt" else ids

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=False):
        text = "".join(f"<|{m['role']}|>{m['content']}" for m in messages) + ("<|assistant|>" if add_generation_prompt else "")
        return self.encode(text) if tokenize else text

    def decode(self, ids, skip_special_tokens=False, **k):
        return " ".join(str(int(i)) for i in ids)



def dit_graph_mode(args):
    """GraphDiT trajectory launch mode from the bench switches: --no-graph -> launches, --graph -> hipGraph replay, else the
    library's choice (launches when the trajectory runs alone, the replay when it overlaps the LLM decode)."""
    return False if getattr(args, "no_graph", False) else (True if getattr(args, "graph", False) else None)

def build_llm(name: str, device, dtype=torch.bfloat16, seed: int = 0, **overrides):
    import transformers
    spec = dict(LLM_CONFIGS[name], **overrides)
    kind = spec.pop("cls")
    cfg_cls = getattr(transformers, kind + "Config")
    model_cls = getattr(transformers, kind + "ForCausalLM")
    cfg = cfg_cls(**spec)
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            model = model_cls(cfg)
    finally:
        torch.set_default_dtype(prev)
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    return model


def build_orchestrator(llm, graph_decoder, device, dtype=torch.bfloat16):
    from .modeling_llamole import GraphLLMForCausalMLM
    tok = SyntheticTokenizer(llm.config.vocab_size)
    enc = types.SimpleNamespace(hidden_size=512)
    pred = types.SimpleNamespace(text_input_size=768, available=None)
    m = GraphLLMForCausalMLM(types.SimpleNamespace(compute_dtype=dtype), types.SimpleNamespace(),
                             types.SimpleNamespace(learned_query_size=8), llm, graph_decoder, pred, enc,
                             dict(tok.special), tok)
    torch.manual_seed(1)
    for name in ("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor"):
        getattr(m, name).to(device=device, dtype=dtype)
    return m, tok


def gen_kwargs(tok, new_tokens: int):
    return dict(do_sample=True, temperature=0.6, top_p=0.9, top_k=50, max_new_tokens=new_tokens,      # top_k: GeneratingArguments default
                eos_token_id=[tok.eos_token_id] + list(tok.special.values()), pad_token_id=tok.pad_token_id)


def _decode_label(mode: str, fused: dict) -> str:
    """What actually runs per decoded token (the label used to say "stock HF forward" while the layers ran fused HIP launches)."""
    if mode == "hf":
        return "HF generate() (stock modules)"
    layers = fused.get("decoder_layers_5_launches")
    body = (f"{layers} decoder layers as five fused HIP launches each (RMSNorm-prologue q|k|v GEMV, rope + KV append + attention, o_proj + "
            "residual, RMSNorm-prologue gate|up GEMV + SiLU*mul, down_proj + residual), one-launch rotary / mask prologue, lm_head GEMV and the "
            "fused sampler, over a static KV cache") if layers else "HF decoder modules over a StaticCache (per-op HIP kernels under nn.Linear where enabled)"
    return body + ("; one hipGraph replayed per token" if mode == "graph" else "; launched eagerly")


def build_e2e_step(args, graph_decoder, device, props, rank: int):
    """Returns (step_fn, info).  step_fn(i) -> list of integer molecule graphs for the rank's batch."""
    llm = build_llm(args.llm, device)
    orch, tok = build_orchestrator(llm, graph_decoder, device)
    n_accel = 0
    fused = {}
    if args.llm_linear == "hip":
        from .llm_accel import accelerate_llm
        fused = accelerate_llm(llm, fuse=bool(args.llm_fuse),
                               layers=args.llm_decode != "hf" and getattr(args, "llm_layer_fuse", True),
                               model_decode=getattr(args, "llm_model_fuse", True))
        n_accel = fused.pop("linears", 0)
    if args.llm_decode != "hf":
        reuse = getattr(args, "query_kv_reuse", True)
        orch.enable_graphed_decode(use_graph=(args.llm_decode == "graph"), fused_cache=bool(fused.get("decode_attention")),
                                   reuse_query_kv=reuse)
        fused["kv_append"] = bool(fused.get("decode_attention"))
        fused["query_forward_reuses_decode_kv"] = bool(reuse)
    B = props.shape[0]
    g = torch.Generator().manual_seed(100 + rank)
    prompt = torch.randint(5, 1000, (B, args.cutoff_len), generator=g).to(device)
    mask = torch.ones_like(prompt)
    kw = gen_kwargs(tok, args.new_tokens)
    n_nodes = torch.full((B,), graph_decoder.max_n_nodes, dtype=torch.int64)
    last = {}

    pipeline = bool(getattr(args, "pipeline", True))
    group = max(1, int(getattr(args, "dit_group", 1))) if pipeline else 1
    pending = {"h": None}
    waiting = []          # (props, cond) of prompts whose trajectory has not been enqueued yet (diffusion batched over `group` prompts)
    dit_ms: list = []

    def collect():
        h, pending["h"] = pending["h"], None
        if h is None:
            return None
        mols, _ = h.result()
        dit_ms.append(h.run_ms)
        return mols

    def launch_group():
        """Enqueue ONE reverse diffusion for the prompts waiting: their conditions form a batch, so the ~10^4 kernel launches of a
        trajectory -- each of which also costs the concurrently decoding LLM stream a dispatch slot -- are shared by `group` molecules."""
        if not waiting:
            return
        g_props = torch.cat([w[0] for w in waiting], dim=0)
        g_cond = torch.cat([w[1] for w in waiting], dim=0)
        k = len(waiting)
        del waiting[:]
        pending["h"] = graph_decoder.generate_graphs_async(g_props, g_cond.float(), -200.0, n_nodes=n_nodes.repeat(k)[: g_props.shape[0]],
                                                           seed=1000 * rank + launch_group.count, use_graph=dit_graph_mode(args))
        launch_group.count += 1

    launch_group.count = 0

    def step_fn(i):
        """One prompt batch.  Pipelined (default): the reverse diffusion of batch i is enqueued on a side stream and
        overlaps the LLM decode of batch i+1 (independent prompts); with ``dit_group`` = k the trajectories of k consecutive
        prompt batches are ONE batched trajectory enqueued after the k-th decode.  Returns the molecules that completed since
        the last call (or None), the rest is collected by ``step_fn.finish()`` inside the timed region."""
        from ._trace import mark
        mark("step: enter")
        torch.manual_seed(1000 * rank + i)
        t0 = time.perf_counter()
        analysis, design_ids, cond = orch.design_hidden(prompt, mask, None, **kw)
        if not pipeline:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        if pipeline:
            waiting.append((props, cond))
            prev = None
            if len(waiting) >= group:
                prev = collect()
                mark("step: previous molecules collected")
                launch_group()
                mark("step: trajectory enqueued")
            last.update(llm_enqueue_s=t1 - t0, new_tokens=int(analysis.shape[1]), **orch.timings)
            return prev
        mols, _ = graph_decoder.generate_graphs(props, cond.float(), -200.0, n_nodes=n_nodes, seed=1000 * rank + i,
                                                use_graph=dit_graph_mode(args))
        t2 = time.perf_counter()
        dit_ms.append(graph_decoder.last_run_ms()[0])
        last.update(llm_s=t1 - t0, graphdit_s=t2 - t1, new_tokens=int(analysis.shape[1]), **orch.timings)
        return mols

    def finish():
        out = collect()
        if waiting:               # a last, smaller group
            launch_group()
            more = collect()
            out = (out or []) + (more or [])
        return out

    step_fn.finish = finish
    step_fn.prompt, step_fn.mask, step_fn.gen_kw = prompt, mask, kw
    step_fn.dit_ms = dit_ms
    step_fn.pipeline = pipeline
    step_fn.group = group

    n_params = sum(p.numel() for p in llm.parameters())
    info = {"llm": args.llm, "llm_params": n_params, "llm_weights": "random-init (no network)", "prompt_len": args.cutoff_len,
            "max_new_tokens": args.new_tokens, "sampling": "temperature 0.6, top_k 50, top_p 0.9",
            "llm_decode": _decode_label(args.llm_decode, fused),
            "llm_linear": ("ll_linear (HIP weight-streaming GEMV) under %d nn.Linear modules for decode-shaped calls" % n_accel)
                          if n_accel else "PyTorch-ROCm default (hipBLASLt)",
            "llm_fused_elementwise": fused,
            "pipeline": (("GraphDiT of prompt batch i runs on a side HIP stream and overlaps the LLM decode of batch i+1" if group == 1 else
                          f"the reverse diffusions of {group} consecutive prompt batches run as ONE batched trajectory on a side HIP stream "
                          f"and overlap the LLM decodes of the next {group}")
                         if pipeline else "none (LLM decode, then GraphDiT, per batch)"),
            "timing_breakdown": last}
    return step_fn, info, orch, llm
