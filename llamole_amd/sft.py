"""SFT plumbing around ``GraphLLMForCausalMLM.forward`` (BASELINE.json configs[4]).

* ``GraphSFTCollator``  -- reference ``src/data/collator.py:DataCollatorForSeqGraph`` (:26-151) without PyG: pads the token
  features, turns ``molecule_ids`` / ``retro_product_ids`` into ``GraphBatch`` objects (first molecule of a row is also its
  design graph), pads ``retro_labels`` with the label pad id.
* ``sft_step``          -- forward, backward, data-parallel gradient all-reduce, optimizer step.  One process per GPU
  (``torch.distributed`` backend "nccl" = RCCL over xGMI); gradients are reduced in a few large flat buckets with a DIRECT
  all-reduce per bucket -- the trainable set of Llamole's SFT (LoRA adapter + three connectors) is tens of MB, i.e. latency-
  bound on point-to-point xGMI links, so bucket count, not ring bandwidth, is what matters.
The LLM forward/backward is stock HuggingFace on PyTorch-ROCm; the graph side of the loss (GIN encoder forward, GIN
predictor forward + reverse sweep w.r.t. the query condition) runs in the HIP engines.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import torch

from .graph_data import GraphBatch

IGNORE_INDEX = -100


class GraphSFTCollator:
    def __init__(self, pad_token_id: int, mol_id_to_graph: Dict[int, Any], label_pad_token_id: int = IGNORE_INDEX,
                 padding_side: str = "right", pad_to_multiple_of: Optional[int] = None):
        self.pad_token_id = pad_token_id
        self.mol_id_to_graph = mol_id_to_graph
        self.label_pad_token_id = label_pad_token_id
        self.padding_side = padding_side
        self.pad_to_multiple_of = pad_to_multiple_of

    def _pad(self, rows: Sequence[Sequence[int]], value: int, length: int) -> torch.Tensor:
        out = torch.full((len(rows), length), value, dtype=torch.long)
        for i, r in enumerate(rows):
            r = list(r)
            if self.padding_side == "right":
                out[i, :len(r)] = torch.tensor(r, dtype=torch.long)
            else:
                out[i, length - len(r):] = torch.tensor(r, dtype=torch.long)
        return out

    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        mol_graphs, design_graphs, retro_graphs = [], [], []
        for f in features:
            for pos, mid in enumerate(f.get("molecule_ids") or []):
                if pos == 0:
                    design_graphs.append(self.mol_id_to_graph[mid])
                if mid != self.label_pad_token_id and mid in self.mol_id_to_graph:
                    mol_graphs.append(self.mol_id_to_graph[mid])
            for mid in f.get("retro_product_ids") or []:
                if mid != self.label_pad_token_id and mid in self.mol_id_to_graph:
                    retro_graphs.append(self.mol_id_to_graph[mid])
        L = max(len(f["input_ids"]) for f in features)
        if self.pad_to_multiple_of:
            L = (L + self.pad_to_multiple_of - 1) // self.pad_to_multiple_of * self.pad_to_multiple_of
        batch: Dict[str, Any] = {
            "input_ids": self._pad([f["input_ids"] for f in features], self.pad_token_id, L),
            "attention_mask": self._pad([f.get("attention_mask", [1] * len(f["input_ids"])) for f in features], 0, L),
        }
        if all("labels" in f for f in features):
            batch["labels"] = self._pad([f["labels"] for f in features], self.label_pad_token_id, L)
        # the reference's preprocessing emits "molecule_properties" (processors/mmsupervised.py:286-310), which its collator hands to
        # tokenizer.pad as is; "property" is the raw dataset column (aligner.py:138), accepted as well
        pkey = "molecule_properties" if "molecule_properties" in features[0] else ("property" if "property" in features[0] else None)
        if pkey is not None:
            batch["molecule_properties"] = torch.tensor([f[pkey] for f in features], dtype=torch.float32)
        batch["molecule_graphs"] = GraphBatch.from_data_list(mol_graphs) if mol_graphs else None
        batch["design_graphs"] = GraphBatch.from_data_list(design_graphs) if design_graphs else None
        batch["retro_product_graphs"] = GraphBatch.from_data_list(retro_graphs) if retro_graphs else None
        labels = [f.get("retro_labels") for f in features]
        if any(r is not None for r in labels):
            R = max(len(r) for r in labels if r is not None)
            batch["retro_labels"] = torch.tensor([(list(r) + [self.label_pad_token_id] * (R - len(r))) if r is not None
                                                  else [self.label_pad_token_id] * R for r in labels], dtype=torch.int64)
        else:
            batch["retro_labels"] = None
        return batch


def to_device(batch: Dict[str, Any], device) -> Dict[str, Any]:
    out = {}
    for k, v in batch.items():
        out[k] = v.to(device) if hasattr(v, "to") and v is not None else v
    return out


def sft_step(model, batch: Dict[str, Any], optimizer=None, bucket_bytes: int = 64 << 20, masters: "Optional[MasterWeights]" = None) -> Dict[str, float]:
    """One SFT step.  Under ``torch.distributed`` every rank calls it on its own shard; gradients are averaged.  ``masters``: the
    optimizer steps the fp32 twins of the trainable parameters (MasterWeights; the optimizer was built on ``masters.masters``)."""
    from .distributed import allreduce_gradients
    out = model(**batch)
    loss = out.loss
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
        for p in (masters.params if masters is not None else ()):
            p.grad = None
    loss.backward()
    params = [p for p in model.parameters() if p.requires_grad]      # fixed list: identical bucket layout on every rank
    allreduce_gradients(params, bucket_bytes=bucket_bytes)
    if optimizer is not None:
        if masters is not None:
            masters.grads_to_masters()
        optimizer.step()
        if masters is not None:
            masters.masters_to_params()
    log = {k: float(v) for k, v in out.additional_log_info.items()}
    log["loss"] = float(loss.detach())
    return log


class MasterWeights:
    """fp32 master copies of the trainable parameters.  The reference casts every trainable parameter -- LoRA A / B and the
    ``modules_to_save`` embedding matrices -- to fp32 (adapter.py:263-265, cast_trainable_params_to_fp32) and lets autocast run the forward
    in bf16; here the model keeps its bf16 tensors (the kernels and fused adapters read those) and the OPTIMIZER steps fp32 twins:
    ``grads_to_masters`` before ``optimizer.step()``, ``masters_to_params`` after it.  At lr 1e-4 an AdamW step on a bf16 weight of
    magnitude 0.02 is about one bf16 ulp, so without the twins most updates of the learned-query rows round away once the cosine decay
    starts.  ``params`` that already are fp32 are stepped in place.  The twin of a parameter is reachable as
    ``param._ll_master`` -- ``lora_state_dict`` saves that (fp32 on disk, as peft does)."""

    def __init__(self, params):
        self.params = list(params)
        self.masters = []
        for p in self.params:
            if p.dtype == torch.float32:
                self.masters.append(p)
            else:
                # a resumed adapter was saved in fp32: start the twin from that value when it is the one the bf16 tensor was rounded from
                src = getattr(p, "_ll_resume_fp32", None)
                if src is None or tuple(src.shape) != tuple(p.shape) or not torch.equal(src.to(device=p.device).to(p.dtype), p.detach()):
                    src = p.detach().float()
                m = torch.nn.Parameter(src.to(device=p.device, dtype=torch.float32).clone(), requires_grad=True)
                p._ll_master = m
                self.masters.append(m)

    def grads_to_masters(self):
        for p, m in zip(self.params, self.masters):
            if m is not p:
                m.grad = None if p.grad is None else p.grad.detach().float()

    def masters_to_params(self):
        with torch.no_grad():
            for p, m in zip(self.params, self.masters):
                if m is not p:
                    p.copy_(m)

    def resync(self):
        """After the bf16 parameters were overwritten from outside (a resumed adapter, a broadcast from rank 0): masters follow."""
        with torch.no_grad():
            for p, m in zip(self.params, self.masters):
                if m is not p:
                    m.copy_(p.float())


def trained_value(p: torch.Tensor) -> torch.Tensor:
    """The value to SAVE for a trainable parameter: its fp32 master when training kept one (MasterWeights), else the parameter."""
    m = getattr(p, "_ll_master", None)
    return (m if m is not None else p).detach()


class _LoRAFunction(torch.autograd.Function):
    """``y = x W^T + b + scale * (x A^T) B^T`` with the rank-r terms folded into GEMM epilogues: the forward adds the adapter term
    to the base product in place (``addmm_``), the reverse sweep does the same for ``dx``.  Written op by op (`base(x) + (x A^T) B^T *
    scale`) every wrapped projection pays four extra elementwise passes over [rows, N] / [rows, K] activations per step -- at 6 x 2048
    tokens through 224 Mistral-7B projections that was ~11 % of the SFT step (profiles/r3_sft_op_by_op_lora_kernel_stats.csv: bf16 add / scalar-mul
    kernels).  One rounding to bf16 instead of three: not further from the f32 result than the op-by-op form."""

    @staticmethod
    def forward(ctx, x, w, bias, a, b, scale):
        x2 = x.reshape(-1, x.shape[-1])
        xa = x2 @ a.t()                                   # [rows, r]
        y = torch.nn.functional.linear(x2, w, bias)       # [rows, N]
        y.addmm_(xa, b.t(), alpha=scale)
        ctx.save_for_backward(x2, xa, w, a, b)
        ctx.scale, ctx.x_shape = scale, x.shape
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, g):
        x2, xa, w, a, b = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        ga = (g2 @ b) * ctx.scale                         # [rows, r]
        db = (g2.t() @ xa) * ctx.scale                    # [N, r]
        da = ga.t() @ x2                                  # [r, K]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = g2 @ w
            dx.addmm_(ga, a)
            dx = dx.view(ctx.x_shape)
        return dx, None, None, da, db, None


class LoRALinear(torch.nn.Module):
    """``y = W x + b + (alpha / r) * B(A x)`` around a frozen ``nn.Linear`` (Hu et al. 2021) -- the adapter form Llamole's
    SFT trains (reference YAMLs: finetuning_type lora, lora_target all); ``peft`` is not in this image, and the decode-time
    stack never sees these modules (the reference merges the adapter before generation, modeling_llamole.py:139-160)."""

    def __init__(self, base: torch.nn.Linear, r: int = 8, alpha: int = 16):
        super().__init__()
        self.base = base
        for p in self.base.parameters():
            p.requires_grad = False
        dev, dt = base.weight.device, base.weight.dtype
        self.lora_a = torch.nn.Parameter(torch.randn(r, base.in_features, device=dev, dtype=dt) * (1.0 / base.in_features) ** 0.5)
        self.lora_b = torch.nn.Parameter(torch.zeros(base.out_features, r, device=dev, dtype=dt))
        self.scale = alpha / r

    def forward(self, x):
        if torch.is_grad_enabled() and x.dtype == self.base.weight.dtype and not torch.is_autocast_enabled():
            return _LoRAFunction.apply(x, self.base.weight, self.base.bias, self.lora_a, self.lora_b, self.scale)
        return self.base(x) + (x @ self.lora_a.t()) @ self.lora_b.t() * self.scale

    def merged_weight(self) -> torch.Tensor:
        return self.base.weight + (self.lora_b @ self.lora_a) * self.scale


def add_lora(model: torch.nn.Module, r: int = 8, alpha: int = 16, targets=("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj",
                                                                           "up_proj", "down_proj")) -> int:
    """Freeze ``model`` and wrap its target Linears with LoRA adapters; returns the number of wrapped modules."""
    for p in model.parameters():
        p.requires_grad = False
    n = 0
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            if name in targets and type(child) is torch.nn.Linear:
                setattr(parent, name, LoRALinear(child, r, alpha))
                n += 1
    return n


def embedding_module_names(model: torch.nn.Module) -> set:
    """Last name components of the input / output embedding modules (reference adapter.py:224-230): {"embed_tokens", "lm_head"} for the
    Llama / Mistral / Qwen2 families."""
    emb = [m for m in (model.get_input_embeddings(), model.get_output_embeddings()) if m is not None]
    return {n.split(".")[-1] for n, m in model.named_modules() if any(m is e for e in emb)}


def _untie_output_embedding(model: torch.nn.Module) -> bool:
    """peft's ModulesToSaveWrapper trains and stores a private copy of each module, so embed_tokens and lm_head stop sharing one tensor
    the moment both are in modules_to_save; same here."""
    inp, out = model.get_input_embeddings(), model.get_output_embeddings()
    if inp is None or out is None or out.weight is not inp.weight:
        return False
    out.weight = torch.nn.Parameter(inp.weight.detach().clone(), requires_grad=inp.weight.requires_grad)
    if getattr(model, "config", None) is not None:
        model.config.tie_word_embeddings = False
    return True


def enable_modules_to_save(model: torch.nn.Module, names) -> List[str]:
    """Make every module whose last name component is in ``names`` fully trainable next to the LoRA adapters (peft ``modules_to_save``;
    the reference puts the resized embed_tokens / lm_head there, adapter.py:224-233).  Returns the module names."""
    names = set(names)
    if not names:
        return []
    if names >= embedding_module_names(model) and len(embedding_module_names(model)) == 2:
        _untie_output_embedding(model)
    hit = []
    for n, mod in model.named_modules():
        if n.split(".")[-1] in names and not isinstance(mod, LoRALinear) and getattr(mod, "weight", None) is not None:
            for p in mod.parameters(recurse=False):
                p.requires_grad = True
            hit.append(n)
    return hit


def merge_lora_adapter(model: torch.nn.Module, adapter_dir: str) -> int:
    """Merge a LoRA adapter stored in peft's on-disk layout into ``model`` in place: ``W += (alpha / r) * B A`` for every
    ``<module>.lora_A/lora_B`` pair, ``modules_to_save`` weights (the resized embed_tokens / lm_head the reference trains,
    adapter.py:224-233) copied over.  What ``PeftModel.from_pretrained(...).merge_and_unload()`` does (reference
    adapter.py:190-192), for images without ``peft``.  Returns the number of merged Linears."""
    import json
    import os
    with open(os.path.join(adapter_dir, "adapter_config.json")) as f:
        cfg = json.load(f)
    if cfg.get("peft_type", "LORA") != "LORA":
        raise ValueError(f"unsupported adapter type {cfg.get('peft_type')}")
    r, alpha = int(cfg["r"]), float(cfg["lora_alpha"])
    scale = alpha / (r ** 0.5) if cfg.get("use_rslora") else alpha / r
    st_path = os.path.join(adapter_dir, "adapter_model.safetensors")
    if os.path.exists(st_path):
        from safetensors.torch import load_file
        tensors = load_file(st_path)
    else:
        tensors = torch.load(os.path.join(adapter_dir, "adapter_model.bin"), map_location="cpu", weights_only=True)
    strip = lambda k: k[len("base_model.model."):] if k.startswith("base_model.model.") else k      # noqa: E731
    saved = {strip(k).replace(".modules_to_save.default", "").replace(".modules_to_save", "").split(".")[-2] for k in tensors if ".lora_" not in k and "." in strip(k)}
    if len(embedding_module_names(model)) == 2 and saved >= embedding_module_names(model):
        _untie_output_embedding(model)     # the adapter trained both matrices separately (peft copies): keep both
    mods = dict(model.named_modules())
    # remove_duplicate=False: with tie_word_embeddings named_parameters() lists the shared tensor once (as embed_tokens), and an adapter
    # whose modules_to_save carries lm_head.weight as well must still find its target
    params = dict(model.named_parameters(remove_duplicate=False))
    merged = 0
    with torch.no_grad():
        for k, a in tensors.items():
            name = strip(k)
            if name.endswith(".lora_A.weight"):
                base = name[:-len(".lora_A.weight")]
                b = tensors[k.replace("lora_A", "lora_B")]
                lin = mods[base]
                delta = (b.float() @ a.float()) * scale
                if cfg.get("fan_in_fan_out"):
                    delta = delta.t()
                lin.weight.add_(delta.to(device=lin.weight.device, dtype=lin.weight.dtype))
                merged += 1
            elif ".lora_" not in name:              # modules_to_save: full replacement weights
                # peft strips the adapter name from saved keys, so a modules_to_save tensor arrives as `<module>.weight` or as
                # `<module>.modules_to_save[.<adapter>].weight`; try the name as it is, then without the wrapper segment
                tgt = params.get(name)
                if tgt is None:
                    tgt = params.get(name.replace(".modules_to_save.default", "").replace(".modules_to_save", ""))
                if tgt is None:
                    raise KeyError(f"adapter tensor {k!r} has no parameter in the base model (modules_to_save target missing)")
                if tuple(tgt.shape) != tuple(a.shape):
                    raise ValueError(f"adapter tensor {k!r} has shape {tuple(a.shape)} but the model's {name!r} is {tuple(tgt.shape)}: "
                                     "resize the embeddings (resize_vocab) before merging the adapter")
                tgt.copy_(a.to(device=tgt.device, dtype=tgt.dtype))
    return merged


def lora_state_dict(model: torch.nn.Module, modules_to_save: Sequence[str] = ()) -> Dict[str, torch.Tensor]:
    """The trained adapter of ``model`` (``LoRALinear`` wrappers of ``add_lora``) under peft's key names:
    ``base_model.model.<module>.lora_A.weight`` [r, in], ``...lora_B.weight`` [out, r]; every module whose last name component is in
    ``modules_to_save`` contributes its full ``<module>.weight`` (the resized embed_tokens / lm_head, reference adapter.py:224-233).
    Tied weights are written once per name, as peft does."""
    out: Dict[str, torch.Tensor] = {}
    for name, mod in model.named_modules():
        if isinstance(mod, LoRALinear):
            out[f"base_model.model.{name}.lora_A.weight"] = trained_value(mod.lora_a).cpu().contiguous()
            out[f"base_model.model.{name}.lora_B.weight"] = trained_value(mod.lora_b).cpu().contiguous()
        elif modules_to_save and name.split(".")[-1] in modules_to_save and getattr(mod, "weight", None) is not None:
            out[f"base_model.model.{name}.weight"] = trained_value(mod.weight).cpu().clone().contiguous()
    return out


def save_lora_adapter(model: torch.nn.Module, adapter_dir: str, modules_to_save: Sequence[str] = (), base_model_name: Optional[str] = None) -> int:
    """Write the adapter ``add_lora`` trained in peft's on-disk layout -- ``adapter_model.safetensors`` + ``adapter_config.json`` -- what
    ``PeftModel.save_pretrained`` leaves in the reference's output_dir (modeling_llamole.py:463-476 with save_peft_format) and what
    ``merge_lora_adapter`` / ``PeftModel.from_pretrained`` read back.  Returns the number of adapted Linears."""
    import json
    import os
    from safetensors.torch import save_file
    wrapped = [(n, m) for n, m in model.named_modules() if isinstance(m, LoRALinear)]
    if not wrapped:
        raise ValueError("save_lora_adapter: the model carries no LoRALinear module (call add_lora first)")
    r = wrapped[0][1].lora_a.shape[0]
    alpha = wrapped[0][1].scale * r
    if any(m.lora_a.shape[0] != r or abs(m.scale * r - alpha) > 1e-6 for _, m in wrapped):
        raise ValueError("save_lora_adapter: mixed ranks / alphas cannot be described by one adapter_config.json")
    os.makedirs(adapter_dir, exist_ok=True)
    save_file(lora_state_dict(model, modules_to_save), os.path.join(adapter_dir, "adapter_model.safetensors"))
    with open(os.path.join(adapter_dir, "adapter_config.json"), "w") as f:
        json.dump({"peft_type": "LORA", "task_type": "CAUSAL_LM", "base_model_name_or_path": base_model_name, "r": int(r),
                   "lora_alpha": float(alpha), "lora_dropout": 0.0, "target_modules": sorted({n.split(".")[-1] for n, _ in wrapped}),
                   "use_rslora": False, "fan_in_fan_out": False, "bias": "none", "inference_mode": True,
                   "modules_to_save": list(modules_to_save) if modules_to_save else None}, f, indent=2)
    return len(wrapped)


def load_lora_adapter(model: torch.nn.Module, adapter_dir: str) -> int:
    """Resume: copy a saved adapter's A / B (and modules_to_save weights) into the ``LoRALinear`` wrappers of ``model`` WITHOUT merging,
    so that training continues on the adapter -- the reference resumes the language-model adapter only (trainer.py:232-234).  Returns
    the number of adapters restored; raises when a saved pair has no wrapper or another shape."""
    import os
    from safetensors.torch import load_file
    tensors = load_file(os.path.join(adapter_dir, "adapter_model.safetensors"))
    saved = {k.split(".")[-2] for k in tensors if ".lora_" not in k and k.count(".") >= 1}
    if len(embedding_module_names(model)) == 2 and saved >= embedding_module_names(model):
        _untie_output_embedding(model)
    mods = dict(model.named_modules())
    params = dict(model.named_parameters(remove_duplicate=False))
    n = 0
    with torch.no_grad():
        for k, v in tensors.items():
            name = k[len("base_model.model."):] if k.startswith("base_model.model.") else k
            if name.endswith(".lora_A.weight") or name.endswith(".lora_B.weight"):
                base, which = name[:-len(".lora_A.weight")], name[-len("lora_A.weight"):-len(".weight")]
                mod = mods.get(base)
                if not isinstance(mod, LoRALinear):
                    raise KeyError(f"adapter tensor {k!r}: {base!r} is not a LoRA-wrapped module of this model")
                tgt = mod.lora_a if which == "lora_A" else mod.lora_b
                if tuple(tgt.shape) != tuple(v.shape):
                    raise ValueError(f"adapter tensor {k!r} has shape {tuple(v.shape)}, the model's adapter {tuple(tgt.shape)}")
                tgt.copy_(v.to(device=tgt.device, dtype=tgt.dtype))
                if v.dtype == torch.float32 and tgt.dtype != torch.float32:
                    tgt._ll_resume_fp32 = v.clone()        # MasterWeights starts the fp32 twin from the saved value, not from its rounding
                n += which == "lora_A"
            else:
                tgt = params.get(name)
                if tgt is None:       # a wrapped module keeps its Linear under `.base`
                    head, _, leaf = name.rpartition(".")
                    tgt = params.get(f"{head}.base.{leaf}")
                if tgt is None:
                    raise KeyError(f"adapter tensor {k!r} has no parameter in the model")
                tgt.copy_(v.to(device=tgt.device, dtype=tgt.dtype))
                if v.dtype == torch.float32 and tgt.dtype != torch.float32:
                    tgt._ll_resume_fp32 = v.clone()
    return n
