"""`main.py train <config.yaml>` counterpart (BASELINE.json configs[4]): the multimodal SFT workflow of
reference ``src/train/mmsft/workflow.py:41-118`` as a thin loop around ``sft.sft_step``.

Reads the keys of ``config/train/mistral_lora.yaml`` (model_name_or_path, new_special_tokens, graph_*_path, graph_lm_connector_path,
finetuning_type lora, lora_target, lora_rank, lora_alpha, learned_query_size, dataset, dataset_dir, cutoff_len, output_dir,
logging_steps, save_steps, per_device_train_batch_size, gradient_accumulation_steps, learning_rate, num_train_epochs, max_steps,
lr_scheduler_type cosine, warmup_ratio, bf16 / pure_bf16, loss_weight_{lm,design,retro}, resume_from_checkpoint / adapter_name_or_path).

What it is: data-parallel SFT, one process per GPU (``torch.distributed`` "nccl" = RCCL over xGMI; bucketed direct all-reduce of the
adapter + connector gradients, ``distributed.allreduce_gradients``), LoRA on the HF language model (``sft.add_lora``; peft is not in this
image), the graph side of the loss on the HIP engines, and an ``output_dir`` the eval driver loads back: adapter in peft's layout,
``connector/*.pt``, ``graphllm_config.json`` (reference modeling_llamole.py:439-519).  Resuming restores the language-model adapter only
(reference trainer.py:232-234) plus the connectors named by ``graph_lm_connector_path``.

What it is not: the HF Trainer (evaluation loop, metrics, plotting, DeepSpeed / FSDP) or LlamaFactory's chat-template library (889 lines,
src/data/template.py) -- the prompt is rendered with the tokenizer's own chat template, as the eval driver does.
The MolQA text -> feature transform restates reference ``src/data/aligner.py:35-96`` (SMILES spans -> ``<molecule>``, body-token
insertion) and ``src/data/processors/mmsupervised.py:139-262`` (label masks, balanced truncation).
"""
from __future__ import annotations

import json
import math
import os
import re
import time
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Tuple

import torch

from .eval import PROPERTY_NAMES, load_dataset_records, load_tokenizer, load_yaml_args
from .sft import IGNORE_INDEX

NO_LABEL_INDEX = -200      # reference extras/constants.py: "no label" for retro steps and absent properties

_MOL = re.compile(r"<mol_start>(.*?)<mol_end>")
_DESIGNED = re.compile(r"(<design_start><design_end>)<mol_start>(.*?)<mol_end>")
_STEP = re.compile(r"(This is step \d+ in the retrosynthesis process\..*?<retro_start>.*?<retro_end>)(.*?)(?=This is step \d+|$)")


def convert_molqa_record(rec: Dict[str, Any], learned_query_size: int) -> Dict[str, Any]:
    """One MolQA training record -> {prompt, response, molecules[SMILES], retro_products[SMILES], retro_labels, property[10]}
    (aligner.py:97-141).  In the response every ``<mol_start>S<mol_end>`` becomes ``<molecule>`` (the designed molecule right after
    ``<design_start><design_end>`` additionally keeps its SMILES inside ``<rollback_start>..<rollback_end>``), ``<design_start>`` gets
    its body tokens, and the i-th retrosynthesis step gets ``<retro_body>`` x q when it has a label."""
    out = rec.get("output") or ""
    smiles = _MOL.findall(out)
    products = [m.strip() for m in re.findall(r"<retro_end>(.*?)>>", out)]
    labels = list(rec.get("retro") or [])
    text = _DESIGNED.sub(lambda m: f"{m.group(1)}<molecule><rollback_start>{m.group(2)}<rollback_end>", out)
    text = _MOL.sub("<molecule>", text)
    text = re.sub(r"<design_start>(.*?)<design_end>", "<design_start>" + "<design_body>" * learned_query_size + "<design_end>", text)
    pieces, last = [], 0
    for i, m in enumerate(_STEP.finditer(text)):
        label = labels[i] if i < len(labels) else None
        pieces.append(text[last:m.start()])
        if label is not None and re.search(r"<retro_start>(.*?)<retro_end>", m.group(1)):
            # the reference drops the step's trailing text when it inserts the bodies (aligner.py:75-79): kept as is
            pieces.append(re.sub(r"<retro_start>.*?<retro_end>", "<retro_start>" + "<retro_body>" * learned_query_size + "<retro_end>", m.group(1)))
        else:
            pieces.append(m.group(1) + m.group(2))
        last = m.end()
    pieces.append(text[last:])
    content = "\n".join(x for x in (rec.get("instruction"), rec.get("input")) if x)
    return {"prompt": content, "response": "".join(pieces), "molecules": smiles, "retro_products": products,
            "retro_labels": [NO_LABEL_INDEX if v is None else int(v) for v in labels],
            "property": [(rec.get("property") or {}).get(p) for p in PROPERTY_NAMES]}


def _budget(source_len: int, target_len: int, cutoff: int) -> Tuple[int, int]:
    """How a (prompt, response) pair shares ``cutoff`` tokens (mmsupervised.py:44-55)."""
    if target_len * 2 < cutoff:
        mx = cutoff
    elif source_len * 2 < cutoff:
        mx = cutoff - source_len
    else:
        mx = int(cutoff * (target_len / (source_len + target_len)))
    t = min(mx, target_len)
    return max(cutoff - t, 0), t


def encode_example(tokenizer, ex: Dict[str, Any], token_id: Dict[str, int], cutoff_len: int, smiles_to_id: Dict[str, int]) -> Dict[str, Any]:
    """Token features of one converted record (mmsupervised.py:139-262): prompt masked out, every special token masked in the labels
    except ``<design_start>`` / ``<retro_start>``, the response cut so that retro tags stay balanced; molecule / product ids and retro
    labels trimmed to what survived the cut."""
    chat = tokenizer.apply_chat_template([{"role": "user", "content": ex["prompt"]}], tokenize=False, add_generation_prompt=True)
    src = tokenizer(chat, add_special_tokens=False)["input_ids"]
    tgt = tokenizer(ex["response"], add_special_tokens=False)["input_ids"] + [tokenizer.eos_token_id]
    sl, tl = _budget(len(src), len(tgt), cutoff_len)
    src = src[:sl]
    rs, re_ = token_id["<retro_start>"], token_id["<retro_end>"]
    starts = [i for i, t in enumerate(tgt) if t == rs]
    ends = [i for i, t in enumerate(tgt) if t == re_]
    if starts and ends:
        keep = -1
        for s, e in zip(starts, ends):
            if e < tl:
                keep = e
            else:
                break
        tl = keep + 1 if keep >= 0 else min(tl, starts[0])
    tgt = tgt[:tl]
    n_mol = tgt.count(token_id["<molecule>"])
    n_retro = tgt.count(re_)
    assert tgt.count(rs) == n_retro, "unbalanced retro tags after truncation"
    masked = {token_id[t] for t in ("<design_start>", "<design_end>", "<design_body>", "<molecule>", "<retro_start>", "<retro_end>", "<retro_body>")}
    kept = {token_id["<retro_start>"], token_id["<design_start>"]}
    labels = [IGNORE_INDEX] * len(src) + [t if (t in kept or t not in masked) else IGNORE_INDEX for t in tgt]
    props = [NO_LABEL_INDEX if v is None else float(v) for v in ex["property"]]
    return {"input_ids": src + tgt, "attention_mask": [1] * (len(src) + len(tgt)), "labels": labels,
            "molecule_ids": [smiles_to_id[s] for s in ex["molecules"][:n_mol]],
            "retro_product_ids": [smiles_to_id[s] for s in ex["retro_products"][:n_retro]],
            "retro_labels": ex["retro_labels"][:n_retro], "molecule_properties": props}


def build_features(records: List[dict], tokenizer, token_id: Dict[str, int], cutoff_len: int, learned_query_size: int, smiles_to_graph):
    """(features, mol_id_to_graph): every distinct SMILES of the dataset gets an id in sorted order (aligner.py:206-213) and its integer
    graph (mmsupervised.py:57-137 via ``smiles_to_graph``)."""
    conv = [convert_molqa_record(r, learned_query_size) for r in records]
    all_smiles = sorted({s for c in conv for s in c["molecules"]} | {s for c in conv for s in c["retro_products"]})
    sid = {s: i for i, s in enumerate(all_smiles)}
    graphs = {}
    for s, i in sid.items():
        g = smiles_to_graph(s)
        if g is None:
            raise ValueError(f"Invalid SMILES string for molecule {i}: {s}")
        graphs[i] = g
    return [encode_example(tokenizer, c, token_id, cutoff_len, sid) for c in conv], graphs


def lora_targets(llm: torch.nn.Module, spec) -> Tuple[str, ...]:
    """``lora_target: all`` = every Linear of the language model except the output head (reference adapter.py:194-199 /
    find_all_linear_modules); otherwise a comma-separated list of module names."""
    if spec in (None, "all", ["all"]):
        names = {n.split(".")[-1] for n, m in llm.named_modules() if type(m) is torch.nn.Linear}
        return tuple(sorted(names - {"lm_head"}))
    return tuple(t.strip() for t in (spec.split(",") if isinstance(spec, str) else spec))


def cosine_lr(step: int, total: int, warmup: int, base: float) -> float:
    if warmup > 0 and step < warmup:
        return base * (step + 1) / warmup
    prog = (step - warmup) / max(1, total - warmup)
    return base * 0.5 * (1.0 + math.cos(math.pi * min(1.0, prog)))


def run_train(config_path: str, overrides: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """Entry used by ``python main.py train cfg.yaml``; returns {"log": [...], "output_dir": ...} (rank 0 also writes
    ``trainer_log.jsonl`` and the checkpoint into ``output_dir``)."""
    import yaml
    from .distributed import shard_range
    from .modeling_llamole import GraphLLMForCausalMLM
    from .sft import GraphSFTCollator, MasterWeights, add_lora, embedding_module_names, enable_modules_to_save, load_lora_adapter, to_device
    with open(config_path) as f:
        cfg = yaml.safe_load(f) or {}
    cfg.update(overrides or {})
    model_args, data_args, training_args, finetuning_args, _ = load_yaml_args(config_path, overrides)
    for k in ("loss_weight_lm", "loss_weight_design", "loss_weight_retro"):
        setattr(finetuning_args, k, float(cfg.get(k, 1)))
    if cfg.get("finetuning_type", "lora") != "lora":
        raise ValueError("only finetuning_type: lora is provided (the reference's shipped training configs)")
    out_dir = cfg.get("output_dir")
    if not out_dir:
        raise ValueError("output_dir is required")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # every rank checks the output directory BEFORE a process group exists: a refusal on rank 0 alone would leave the others waiting in
    # their first collective
    if os.path.isdir(out_dir) and os.listdir(out_dir) and not cfg.get("overwrite_output_dir", False) and not cfg.get("resume_from_checkpoint"):
        raise ValueError(f"Output directory ({out_dir}) already exists and is not empty. Use overwrite_output_dir to overcome.")
    if world > 1:
        import torch.distributed as dist
        n_dev = torch.cuda.device_count()
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if local >= n_dev and os.environ.get("LLAMOLE_BENCH_SHARED_GPU") != "1":
            raise RuntimeError(f"rank {rank} needs GPU {local}, this node shows {n_dev}")
        torch.cuda.set_device(local % n_dev)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group(os.environ.get("LLAMOLE_DIST_BACKEND", "nccl"))
    # `seed` fixes every random draw of the run on every rank -- the resized embedding rows, the connector and LoRA-A initialisations, the
    # data permutation -- (the reference: transformers.set_seed(training_args.seed) in the Trainer); the replicas are made identical
    # below by a broadcast from rank 0 whatever the RNG streams did
    seed = int(cfg.get("seed", 42))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    tokenizer = load_tokenizer(model_args)
    tokenizer.padding_side = "right"
    model = GraphLLMForCausalMLM.from_pretrained(tokenizer, model_args, data_args, training_args, finetuning_args, load_adapter=False)
    dev = model.device
    targets = lora_targets(model.language_model, cfg.get("lora_target", "all"))
    r = int(cfg.get("lora_rank", 8))
    n_lora = add_lora(model.language_model, r=r, alpha=int(cfg.get("lora_alpha", 2 * r)), targets=targets)
    # modules trained in full next to the adapters (peft modules_to_save).  When special tokens were added to the tokenizer and
    # additional_target is unset the reference puts the input and output embeddings there (adapter.py:224-233): the rows of
    # <design_body> / <retro_body> ARE the learned queries, and <design_start> / <retro_start> stay in the labels
    if cfg.get("additional_target"):
        at = cfg["additional_target"]
        extra_modules = {t.strip() for t in (at.split(",") if isinstance(at, str) else at) if t.strip()}
    elif getattr(model_args, "resize_vocab", False):
        extra_modules = embedding_module_names(model.language_model)
    else:
        extra_modules = set()
    trained_modules = enable_modules_to_save(model.language_model, extra_modules)
    resume = cfg.get("resume_from_checkpoint") or (model_args.adapter_name_or_path or [None])[0]
    if resume:
        load_lora_adapter(model.language_model, resume)
        cpath = model_args.graph_lm_connector_path or os.path.join(resume, "connector")
        for name in ("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor"):
            fp = os.path.join(cpath, name + ".pt")
            if os.path.exists(fp):
                getattr(model, name).load_state_dict(torch.load(fp, map_location=dev, weights_only=True))
    for name in ("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor"):
        for p in getattr(model, name).parameters():
            p.requires_grad = True
    model.train()
    records = load_dataset_records(data_args)
    feats, graphs = build_features(records, tokenizer, model.token_id_dict, data_args.cutoff_len, data_args.learned_query_size, model.smiles_to_graph)
    collate = GraphSFTCollator(tokenizer.pad_token_id, graphs, pad_to_multiple_of=8)
    bs, accum = int(cfg.get("per_device_train_batch_size", 1)), int(cfg.get("gradient_accumulation_steps", 1))
    per_step = bs * accum * world
    steps_per_epoch = max(1, len(feats) // per_step)
    total = int(cfg.get("max_steps", 0)) or int(math.ceil(float(cfg.get("num_train_epochs", 1.0)) * steps_per_epoch))
    warmup = int(cfg.get("warmup_steps", 0)) or int(float(cfg.get("warmup_ratio", 0.0)) * total)
    params = [p for p in model.parameters() if p.requires_grad]
    if world > 1:      # identical replicas whatever each process drew: only rank 0's copy is saved, and only gradients are averaged
        import torch.distributed as dist
        for p in params:
            dist.broadcast(p.data, src=0)
    lr = float(cfg.get("learning_rate", 1e-4))
    # every trainable parameter is optimised in fp32 (reference adapter.py:263-265 casts them; here: fp32 twins of the bf16 tensors)
    masters = MasterWeights(params)
    opt = torch.optim.AdamW(masters.masters, lr=lr, weight_decay=float(cfg.get("weight_decay", 0.0)))
    gen = torch.Generator().manual_seed(seed)
    log: List[Dict[str, Any]] = []
    logging_steps, save_steps = int(cfg.get("logging_steps", 10)), int(cfg.get("save_steps", 0))
    order: List[int] = []

    def save(where):
        if rank == 0:
            model.save_pretrained(where, modules_to_save=tuple(sorted(extra_modules)))
            tokenizer.save_pretrained(where)

    t0 = time.perf_counter()
    for step in range(total):
        for g in opt.param_groups:
            g["lr"] = cosine_lr(step, total, warmup, lr) if cfg.get("lr_scheduler_type", "cosine") == "cosine" else lr
        opt.zero_grad(set_to_none=True)
        for p in params:
            p.grad = None
        agg: Dict[str, float] = {}
        for micro in range(accum):
            if len(order) < bs * world:          # a new epoch: the same permutation on every rank (same seed), each takes its shard
                order += torch.randperm(len(feats), generator=gen).tolist()
            take, order = order[:bs * world], order[bs * world:]
            mine = [take[i] for i in shard_range(len(take), rank, world)]
            batch = to_device(collate([feats[i] for i in mine]), dev)
            last = micro == accum - 1
            out = _micro_step(model, batch, params, accum, reduce=last)
            for k, v in out.items():
                agg[k] = agg.get(k, 0.0) + v / accum
        masters.grads_to_masters()
        if cfg.get("max_grad_norm", 1.0):
            torch.nn.utils.clip_grad_norm_(masters.masters, float(cfg.get("max_grad_norm", 1.0)))
        opt.step()
        masters.masters_to_params()
        agg.update(step=step + 1, lr=opt.param_groups[0]["lr"], elapsed_s=time.perf_counter() - t0)
        log.append(agg)
        if rank == 0 and ((step + 1) % logging_steps == 0 or step + 1 == total):
            print(json.dumps(agg), flush=True)
        if save_steps and (step + 1) % save_steps == 0 and step + 1 < total:
            save(os.path.join(out_dir, f"checkpoint-{step + 1}"))
    save(out_dir)
    if rank == 0:
        with open(os.path.join(out_dir, "trainer_log.jsonl"), "w") as f:
            for row in log:
                f.write(json.dumps(row) + "\n")
        with open(os.path.join(out_dir, "train_results.json"), "w") as f:
            json.dump({"train_steps": total, "train_runtime": time.perf_counter() - t0, "train_loss": sum(r["loss"] for r in log) / max(1, len(log)),
                       "lora_modules": n_lora, "lora_targets": list(targets), "modules_to_save": trained_modules, "world_size": world}, f, indent=1)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    return {"log": log, "output_dir": out_dir, "lora_modules": n_lora, "modules_to_save": trained_modules}


def _micro_step(model, batch, params, accum: int, reduce: bool) -> Dict[str, float]:
    """Forward + backward of one micro batch (loss / accum); the gradient all-reduce runs once, after the last micro batch."""
    from .distributed import allreduce_gradients
    out = model(**batch)
    (out.loss / accum).backward()
    if reduce:
        allreduce_gradients(params)
    log = {k: float(v) for k, v in out.additional_log_info.items()}
    log["loss"] = float(out.loss.detach())
    return log
