"""`main.py eval <config.yaml>` counterpart: the two-phase MolQA evaluation driver of
reference ``src/eval/workflow.py:64-219`` + ``src/eval/dataset.py:26-78`` on the MI355X engines.

Keeps the reference's YAML surface (``config/generate/*.yaml``: model_name_or_path, new_special_tokens,
graph_{decoder,encoder,predictor}_path, adapter_name_or_path, graph_lm_connector_path, max_new_tokens, temperature,
top_p, learned_query_size, dataset, cutoff_len, bf16/pure_bf16, per_device_eval_batch_size) and the result-record
fields (qa_idx, instruction, input, llm_response, response_design, llm_smiles, property, llm_reactions[{reaction,
template, cost}], response_retro).  Differences, all deliberate: no ``raise 'stop'`` / dataset-name gate (the reference
driver does not run as shipped, workflow.py:50-56); prompts shard across ranks (one process per GPU) with one
all-gather of the result records; throughput (molecules/s, GraphDiT steps/s) is reported.
"""
from __future__ import annotations

import json
import math
import os
import re
import time
from types import SimpleNamespace
from typing import Any, Dict, List, Optional

import torch

PROPERTY_NAMES = ["BBBP", "HIV", "BACE", "CO2", "N2", "O2", "FFV", "TC", "SC", "SA"]   # eval/dataset.py:36-47


def remove_extra_spaces(text: str) -> str:
    return re.sub(r"\s+", " ", text).strip()


def load_yaml_args(path: str, overrides: Optional[Dict[str, Any]] = None):
    """YAML -> (model_args, data_args, training_args, finetuning_args, generating_args) namespaces with the attribute
    names the reference dataclasses expose (hparams/parser.py:137-319) for the keys of the generate configs."""
    import yaml
    with open(path, "r") as f:
        cfg = yaml.safe_load(f) or {}
    cfg.update(overrides or {})
    bf16 = bool(cfg.get("bf16", False) or cfg.get("pure_bf16", False))
    fp16 = bool(cfg.get("fp16", False))
    tokens = cfg.get("new_special_tokens")
    if isinstance(tokens, str):
        tokens = [t.strip() for t in tokens.split(",") if t.strip()]
    adapter = cfg.get("adapter_name_or_path")
    if isinstance(adapter, str):
        adapter = [a.strip() for a in adapter.split(",")]
    model_args = SimpleNamespace(
        model_name_or_path=cfg.get("model_name_or_path"), new_special_tokens=tokens, adapter_name_or_path=adapter,
        graph_decoder_path=cfg.get("graph_decoder_path"), graph_encoder_path=cfg.get("graph_encoder_path"),
        graph_predictor_path=cfg.get("graph_predictor_path"), graph_lm_connector_path=cfg.get("graph_lm_connector_path"),
        compute_dtype=torch.bfloat16 if bf16 else (torch.float16 if fp16 else torch.float32),
        disable_graph_model_gradient=cfg.get("disable_graph_model_gradient", True),
        flash_attn=cfg.get("flash_attn", "auto"))
    data_args = SimpleNamespace(dataset=cfg.get("dataset"), dataset_dir=cfg.get("dataset_dir", "data"),
                                template=cfg.get("template"), cutoff_len=int(cfg.get("cutoff_len", 1024)),
                                learned_query_size=int(cfg.get("learned_query_size", 8)),
                                # A* budget of phase 2 (constants of the reference driver, eval/workflow.py:171-173); overridable
                                retro_iterations=int(cfg.get("retro_iterations", 100)),
                                retro_max_planning_time=float(cfg.get("retro_max_planning_time", 30)),
                                expansion_topk=int(cfg.get("expansion_topk", 50)),
                                dynamic_sharding=bool(cfg.get("dynamic_sharding", True)))
    training_args = SimpleNamespace(per_device_eval_batch_size=int(cfg.get("per_device_eval_batch_size", 8)),
                                    do_train=bool(cfg.get("do_train", False)), output_dir=cfg.get("output_dir"))
    finetuning_args = SimpleNamespace(finetuning_type=cfg.get("finetuning_type", "lora"))
    gen = {"do_sample": cfg.get("do_sample", True), "temperature": cfg.get("temperature", 0.95), "top_p": cfg.get("top_p", 0.7),
           "top_k": cfg.get("top_k", 50), "num_beams": cfg.get("num_beams", 1), "max_new_tokens": cfg.get("max_new_tokens", 1024),
           "repetition_penalty": cfg.get("repetition_penalty", 1.0), "length_penalty": cfg.get("length_penalty", 1.0)}
    generating_args = SimpleNamespace(**gen, to_dict=lambda: dict(gen))
    return model_args, data_args, training_args, finetuning_args, generating_args


def load_dataset_records(data_args) -> List[dict]:
    info_path = os.path.join(data_args.dataset_dir, "dataset_info.json")
    with open(info_path, "r") as f:
        info = json.load(f)
    name = data_args.dataset.strip()
    if name not in info:
        raise ValueError(f"Dataset {name} not found in dataset_info.json")
    with open(os.path.join(data_args.dataset_dir, info[name]["file_name"]), "r") as f:
        return json.load(f)


def encode_batch(tokenizer, records: List[dict], max_len: int):
    """Chat-templated, max-length (left-)padded prompts + 10-slot property tensor (eval/dataset.py:35-78)."""
    ids, masks, props = [], [], []
    for item in records:
        chat = tokenizer.apply_chat_template([{"role": "user", "content": f"{item['instruction']}\n{item['input']}"}],
                                             tokenize=False, add_generation_prompt=True)
        enc = tokenizer(chat, return_tensors="pt", padding="max_length", truncation=True, max_length=max_len)
        ids.append(enc.input_ids.reshape(-1))
        masks.append(enc.attention_mask.reshape(-1))
        props.append(torch.tensor([item.get("property", {}).get(p, float("nan")) for p in PROPERTY_NAMES], dtype=torch.float32))
    return torch.stack(ids), torch.stack(masks), torch.stack(props)


def run_molqa(model, tokenizer, records: List[dict], cutoff_len: int, batch_size: int, gen_kwargs: Dict[str, Any],
              do_retrosynthesis: bool = True, rank: int = 0, world: int = 1, expansion_topk: int = 50, iterations: int = 100,
              max_planning_time: int = 30, work_queue=None) -> Dict[str, Any]:
    """Phase 1 (design) for this rank's prompts, then phase 2 (retrosynthesis) for the same prompts; returns {"results": [...],
    "stats": {...}}.  Prompts are a static contiguous shard per rank, or -- with ``work_queue`` (distributed.WorkQueue over the
    batches) -- claimed batch by batch from a shared counter, which evens out prompts whose decodes / searches run long."""
    from .distributed import shard_range
    def claim():
        """This rank's prompt batches, one at a time (a batch is claimed only when the previous one is done)."""
        if work_queue is not None:
            for b in work_queue:
                yield list(range(b * batch_size, min((b + 1) * batch_size, len(records))))
        else:
            shard = list(shard_range(len(records), rank, world))
            for lo in range(0, len(shard), batch_size):
                yield shard[lo:lo + batch_size]
    results: List[dict] = []
    taken: List[List[int]] = []          # the batches this rank processed, in order (phase 2 revisits them)
    dev = model.device
    t0 = time.perf_counter()
    dit_ms, dit_steps = 0.0, 0
    for idxs in claim():
        taken.append(idxs)
        ids, mask, props = encode_batch(tokenizer, [records[i] for i in idxs], cutoff_len)
        info = model.generate(input_ids=ids.to(dev), attention_mask=mask.to(dev), molecule_properties=props.to(dev),
                              do_molecular_design=True, do_retrosynthesis=False, rollback=True, **gen_kwargs)
        if work_queue is not None:
            work_queue.beat()
        try:
            ms, st = model.graph_decoder.last_run_ms()
            dit_ms, dit_steps = dit_ms + ms, dit_steps + st
        except Exception:
            pass
        for j, qi in enumerate(idxs):
            text = "".join(info["text_lists"][j])
            rec = {"qa_idx": qi, "instruction": records[qi]["instruction"], "input": records[qi]["input"],
                   "llm_response": text, "response_design": remove_extra_spaces(text), "llm_smiles": info["smiles_list"][j],
                   "property": {n: float(v) for n, v in zip(PROPERTY_NAMES, props[j].tolist()) if not math.isnan(v)}}
            results.append(rec)
    t1 = time.perf_counter()
    if do_retrosynthesis:
        lo = 0
        for idxs in taken:
            ids, mask, _ = encode_batch(tokenizer, [records[i] for i in idxs], cutoff_len)
            smiles = [results[lo + j]["llm_smiles"] for j in range(len(idxs))]
            info = model.generate(input_ids=ids.to(dev), attention_mask=mask.to(dev), do_molecular_design=False,
                                  do_retrosynthesis=True, input_smiles_list=smiles, expansion_topk=expansion_topk,
                                  iterations=iterations, max_planning_time=max_planning_time, **gen_kwargs)
            if work_queue is not None:
                work_queue.beat()
            for j in range(len(idxs)):
                rec = results[lo + j]
                plan = info["retro_plan_dict"][rec["llm_smiles"]]
                rec["llm_reactions"] = ([{"reaction": r, "template": t, "cost": c} for r, t, c in
                                         zip(plan["reaction_list"], plan["templates"], plan["cost"])] if plan["success"] else [])
                new_text = "".join(x for x in info["text_lists"][j] if x is not None)
                rec["llm_response"] = remove_extra_spaces(rec["llm_response"] + new_text)
                rec["response_retro"] = remove_extra_spaces(new_text)
            lo += len(idxs)
    t2 = time.perf_counter()
    n_mine = sum(len(b) for b in taken)
    stats = {"n_prompts": n_mine, "design_s": t1 - t0, "retro_s": t2 - t1,
             "molecules_per_s": n_mine / max(t1 - t0, 1e-9),
             "denoise_steps_per_s": (1e3 * dit_steps / dit_ms) if dit_ms > 0 else None}
    from .distributed import force_dist
    if world > 1 or (force_dist() and torch.distributed.is_initialized()):
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, results)
        results = sorted((r for part in gathered for r in part), key=lambda r: r["qa_idx"])
    return {"results": results, "stats": stats}


def load_tokenizer(model_args):
    """reference loader.py:88-138 in generate mode: left padding, the new special tokens ADDED to the existing additional
    special tokens, pad = eos (eval/workflow.py:77)."""
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(model_args.model_name_or_path, padding_side="left")
    new = list(model_args.new_special_tokens or [])
    before = len(tokenizer)
    if new:
        try:
            tokenizer.add_special_tokens({"additional_special_tokens": new}, replace_additional_special_tokens=False)
        except TypeError:       # transformers >= 5 renamed the keyword
            tokenizer.add_tokens(new, special_tokens=True)
    # the reference flips model_args.resize_vocab when tokens were added (loader.py:121-126); the SFT driver reads it to decide whether the
    # embedding matrices are trained and saved with the adapter (adapter.py:224-233)
    model_args.resize_vocab = bool(getattr(model_args, "resize_vocab", False)) or len(tokenizer) > before
    tokenizer.pad_token = tokenizer.eos_token
    return tokenizer


def special_token_ids(tokenizer, new_special_tokens) -> List[int]:
    """Every added special token stops generation (eval/workflow.py:93-98: eos + tokenizer.additional_special_tokens_ids)."""
    ids = list(getattr(tokenizer, "additional_special_tokens_ids", None) or [])
    extra = []
    for attr in ("additional_special_tokens", "extra_special_tokens"):          # the attribute was renamed in transformers 5
        v = getattr(tokenizer, attr, None)
        extra += list(v.values() if isinstance(v, dict) else (v or []))
    for t in extra + list(new_special_tokens or []):
        i = tokenizer.convert_tokens_to_ids(t)
        if isinstance(i, int) and i >= 0 and i not in ids:
            ids.append(i)
    return ids


def run_eval(config_path: str, overrides: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """Entry used by ``python main.py eval cfg.yaml``."""
    from .modeling_llamole import GraphLLMForCausalMLM
    model_args, data_args, training_args, finetuning_args, generating_args = load_yaml_args(config_path, overrides)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    from .distributed import force_dist
    if world > 1 or force_dist():
        import torch.distributed as dist
        # one process per GPU; LLAMOLE_BENCH_SHARED_GPU=1 / LLAMOLE_DIST_BACKEND=gloo are the single-GPU dry-run switches of bench.py (tests)
        n_dev = torch.cuda.device_count()
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if local >= n_dev and os.environ.get("LLAMOLE_BENCH_SHARED_GPU") != "1":
            raise RuntimeError(f"rank {rank} needs GPU {local}, this node shows {n_dev}")
        torch.cuda.set_device(local % n_dev)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("LLAMOLE_DIST_BACKEND", "nccl")
        if not dist.is_initialized():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local % n_dev))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
    tokenizer = load_tokenizer(model_args)
    gen_kwargs = generating_args.to_dict()
    gen_kwargs["eos_token_id"] = [tokenizer.eos_token_id] + special_token_ids(tokenizer, model_args.new_special_tokens)
    gen_kwargs["pad_token_id"] = tokenizer.pad_token_id
    # the reference always passes load_adapter=True (eval/workflow.py:100-108) and therefore needs an adapter; a YAML without
    # adapter_name_or_path evaluates the base model with the connectors of graph_lm_connector_path here instead of raising
    model = GraphLLMForCausalMLM.from_pretrained(tokenizer, model_args, data_args, training_args, finetuning_args,
                                                 load_adapter=bool(model_args.adapter_name_or_path))
    model.eval()
    accel = model.enable_mi355x_decode()
    model.batch_retro = bool(getattr(generating_args, "batch_retro", False) or (overrides or {}).get("batch_retro", False))
    if rank == 0:
        print(json.dumps({"llm_acceleration": accel}))
    records = load_dataset_records(data_args)
    queue = None
    if world > 1 and bool((overrides or {}).get("dynamic_sharding", getattr(data_args, "dynamic_sharding", True))):
        from .distributed import WorkQueue      # prompts are claimed batch by batch: long decodes / searches do not idle other ranks
        bs = training_args.per_device_eval_batch_size
        queue = WorkQueue((len(records) + bs - 1) // bs, rank, world)
    out = run_molqa(model, tokenizer, records, data_args.cutoff_len,
                    training_args.per_device_eval_batch_size, gen_kwargs, rank=rank, world=world, work_queue=queue,
                    expansion_topk=data_args.expansion_topk, iterations=data_args.retro_iterations,
                    max_planning_time=data_args.retro_max_planning_time)
    from .graph_encoder import check_graph_errors
    check_graph_errors(wait=True)          # nothing malformed may leave the run unreported (the flag of the last GIN call)
    if rank == 0:
        print(json.dumps(out["stats"]))
        if training_args.output_dir:
            os.makedirs(training_args.output_dir, exist_ok=True)
            with open(os.path.join(training_args.output_dir, "molqa_results.json"), "w") as f:
                json.dump(out["results"], f, indent=1)
    return out
