"""Synthetic configs, weights and inputs for the GraphDiT / GIN hot path.

The reference ships neither pretrained weights nor the GraphDiT/GIN hyper-parameter
files (they are downloaded at first run: reference src/model/loader.py:226-231,
267-274, 326-331), so every test, golden fixture and benchmark in this repo uses
seeded synthetic stand-ins of the same *shape*.  Everything here is generated with
``numpy.random.RandomState`` (a frozen bit-stream), so the golden generator that runs
next to the reference and the tests that run on the GPU box rebuild byte-identical
weights and inputs without committing megabytes of tensors.

State-dict key names / shapes follow the reference modules:
  denoiser  : src/model/graph_decoder/transformer.py:24-108, layers.py:23-116,
              conditions.py:19-123
  encoder   : src/model/graph_encoder/model.py:87-205
  predictor : src/model/graph_predictor/model.py:231-304, 394-423
"""
from __future__ import annotations

import json
import math
import os
import re
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

XDIM = 16   # active atom classes   (reference diffusion_utils.py:58-59)
EDIM = 5    # bond classes
YDIM = 10   # property slots
TEXT_DIM = 768
N_ATOM_TYPES = 118

# 10-slot property order used by the reference eval dataset (src/eval/dataset.py:36-47)
PROPERTY_ORDER = ["BBBP", "HIV", "BACE", "CO2", "N2", "O2", "FFV", "TC", "SC", "SA"]

_ACTIVE_SYMBOLS = ["C", "N", "O", "F", "P", "S", "Cl", "Br", "I", "Si", "B", "Se",
                   "Na", "Ge", "Sn", "*"]


# --------------------------------------------------------------------------- GraphDiT
def make_data_meta(max_node: int = 32, seed: int = 0, fixed_n_nodes: Optional[int] = None) -> dict:
    """Synthetic ``data.meta.json`` (keys read by reference diffusion_utils.py:29-59)."""
    rs = np.random.RandomState(seed)
    active_pos = np.sort(rs.choice(N_ATOM_TYPES, XDIM, replace=False))
    atom_type_dist = np.zeros(N_ATOM_TYPES, dtype=np.float64)
    atom_type_dist[active_pos] = rs.uniform(0.1, 1.1, XDIM)
    hist = np.zeros(max_node + 1, dtype=np.float64)
    if fixed_n_nodes is not None:
        hist[fixed_n_nodes] = 1.0
    else:
        lo = min(5, max_node)
        hist[lo:] = rs.uniform(0.5, 1.5, max_node + 1 - lo)
    transition_E = rs.uniform(0.01, 1.01, (N_ATOM_TYPES, N_ATOM_TYPES, EDIM))
    valencies = rs.uniform(0.0, 1.0, 3 * max_node - 2)
    return {
        "active_atoms": list(_ACTIVE_SYMBOLS),
        "max_node": int(max_node),
        "n_atoms_per_mol_dist": [float(v) for v in hist],
        "bond_type_dist": [0.9, 0.06, 0.02, 0.005, 0.015],
        "transition_E": np.round(transition_E, 6).tolist(),
        "atom_type_dist": [float(np.round(v, 6)) for v in atom_type_dist],
        "valencies": [float(np.round(v, 6)) for v in valencies],
    }


def make_dit_config(hidden_size=128, depth=2, num_heads=4, diffusion_steps=50,
                    guide_scale=2.0, mlp_ratio=4.0) -> dict:
    """Synthetic GraphDiT ``config.yaml`` (keys read by reference diffusion_model.py:36-76)."""
    return {
        "diffusion_steps": int(diffusion_steps),
        "guide_scale": float(guide_scale),
        "hidden_size": int(hidden_size),
        "depth": int(depth),
        "num_heads": int(num_heads),
        "mlp_ratio": float(mlp_ratio),
        "drop_condition": 0.1,
        "lambda_train": [1, 10],
        "diffusion_noise_schedule": "cosine",
    }


def dit_weight_shapes(cfg: dict, max_node: int) -> "OrderedDict[str, tuple]":
    """Key -> shape of the reference ``Transformer`` state dict."""
    H, L, heads = cfg["hidden_size"], cfg["depth"], cfg["num_heads"]
    hd = H // heads
    F = XDIM + max_node * EDIM
    Hm = int(H * cfg.get("mlp_ratio", 4.0))
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["x_embedder.0.weight"] = (H, F)
    s["x_embedder.1.weight"] = (H,)
    s["x_embedder.1.bias"] = (H,)
    s["t_embedder.mlp.0.weight"] = (H, 256)
    s["t_embedder.mlp.0.bias"] = (H,)
    s["t_embedder.mlp.2.weight"] = (H, H)
    s["t_embedder.mlp.2.bias"] = (H,)
    s["y_embedder.embedding_drop.weight"] = (YDIM, H)
    for d in range(YDIM):
        s[f"y_embedder.mlps.{d}.0.weight"] = (H, 1)
        s[f"y_embedder.mlps.{d}.0.bias"] = (H,)
        s[f"y_embedder.mlps.{d}.2.weight"] = (H, H)
    s["txt_embedder.embedding_drop.weight"] = (1, H)
    s["txt_embedder.linear.weight"] = (H, TEXT_DIM)
    s["txt_embedder.linear.bias"] = (H,)
    for i in range(L):
        p = f"blocks.{i}."
        s[p + "attn.qkv.weight"] = (3 * H, H)
        s[p + "attn.q_norm.weight"] = (hd,)
        s[p + "attn.q_norm.bias"] = (hd,)
        s[p + "attn.k_norm.weight"] = (hd,)
        s[p + "attn.k_norm.bias"] = (hd,)
        s[p + "attn.proj.weight"] = (H, H)
        s[p + "attn.proj.bias"] = (H,)
        s[p + "mlp.fc1.weight"] = (Hm, H)
        s[p + "mlp.fc1.bias"] = (Hm,)
        s[p + "mlp.fc2.weight"] = (H, Hm)
        s[p + "mlp.fc2.bias"] = (H,)
        s[p + "adaLN_modulation.0.weight"] = (H, H)
        s[p + "adaLN_modulation.0.bias"] = (H,)
        s[p + "adaLN_modulation.2.weight"] = (6 * H, H)
        s[p + "adaLN_modulation.2.bias"] = (6 * H,)
    s["output_layer.xedecoder.fc1.weight"] = (H, H)
    s["output_layer.xedecoder.fc1.bias"] = (H,)
    s["output_layer.xedecoder.fc2.weight"] = (F, H)
    s["output_layer.xedecoder.fc2.bias"] = (F,)
    s["output_layer.adaLN_modulation.0.weight"] = (H, H)
    s["output_layer.adaLN_modulation.0.bias"] = (H,)
    s["output_layer.adaLN_modulation.2.weight"] = (2 * F, H)
    s["output_layer.adaLN_modulation.2.bias"] = (2 * F,)
    return s


def _fill(rs: np.random.RandomState, name: str, shape: tuple) -> np.ndarray:
    """Seeded init: ~xavier weights, small biases, LayerNorm gains near 1.

    Every adaLN layer is randomised (the reference zero-inits the first adaLN Linear,
    transformer.py:74-84, which would make every block an identity and the goldens
    vacuous)."""
    n = int(np.prod(shape))
    is_norm_gain = re.search(r'(norm\d*\.weight|norms\.\d+\.weight|\.1\.weight)$', name) is not None
    if len(shape) == 1 and is_norm_gain:
        return (1.0 + 0.1 * rs.standard_normal(n)).astype(np.float32).reshape(shape)
    if len(shape) == 1:
        return (0.05 * rs.standard_normal(n)).astype(np.float32).reshape(shape)
    if "embedding" in name or "encoder.weight" in name or "text_dropping" in name:
        return (0.5 * rs.standard_normal(n)).astype(np.float32).reshape(shape)
    fan_in = shape[1]
    fan_out = shape[0]
    std = math.sqrt(2.0 / (fan_in + fan_out))
    return (std * rs.standard_normal(n)).astype(np.float32).reshape(shape)


def make_dit_weights(cfg: dict, max_node: int, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    rs = np.random.RandomState(1000 + seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in dit_weight_shapes(cfg, max_node).items():
        sd[k] = torch.from_numpy(_fill(rs, k, shp))
    return sd


def write_dit_dir(path: str, cfg: dict, meta: dict, sd: Dict[str, torch.Tensor]) -> str:
    """Lay out a GraphDiT checkpoint directory as the reference loader expects
    (loader.py:222-246: ``config.yaml``, ``data.meta.json``, ``model.pt``)."""
    import yaml
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    with open(os.path.join(path, "data.meta.json"), "w") as f:
        json.dump(meta, f)
    torch.save(OrderedDict((k, v.clone()) for k, v in sd.items()), os.path.join(path, "model.pt"))
    return path


def make_dit_inputs(batch: int, seed: int = 0, max_node: int = 32,
                    n_nodes_fixed: Optional[int] = None):
    """Synthetic conditions: 10-slot property vectors with NaN for absent slots
    (value ranges from reference data/property_ranges.json) and N(0,1) text embeddings."""
    rs = np.random.RandomState(2000 + seed)
    lo = np.array([0, 0, 0, 0.9, 0.0, 0.0, 0.32, 0.11, 1.0, 1.0])
    hi = np.array([1, 1, 1, 1019.3, 16.7, 122.9, 0.47, 0.55, 5.0, 8.5])
    props = lo + (hi - lo) * rs.uniform(size=(batch, YDIM))
    props[:, :3] = np.round(props[:, :3])
    absent = rs.uniform(size=(batch, YDIM)) < 0.4
    props[absent] = np.nan
    text = rs.standard_normal((batch, TEXT_DIM))
    if n_nodes_fixed is not None:
        n_nodes = np.full(batch, n_nodes_fixed, dtype=np.int64)
    else:
        n_nodes = rs.randint(min(5, max_node), max_node + 1, size=batch).astype(np.int64)
    return (torch.from_numpy(props.astype(np.float32)),
            torch.from_numpy(text.astype(np.float32)),
            torch.from_numpy(n_nodes))


def exp_noise(seed: int, step: int, batch: int, n: int):
    """Exp(1) race noise for one sampling step, in the reference's RNG-order contract:
    X noise ``[B*N,16]`` first, then E noise ``[B*N*N,5]`` (diffusion_utils.py:376-413)."""
    rs = np.random.RandomState((7919 * (seed + 1) + step) % (2 ** 31 - 1))
    qx = rs.exponential(size=(batch * n, XDIM)).astype(np.float32)
    qe = rs.exponential(size=(batch * n * n, EDIM)).astype(np.float32)
    return torch.from_numpy(qx), torch.from_numpy(qe)


# --------------------------------------------------------------------------- GIN
def gin_weight_shapes(num_layer: int, H: int, kind: str, out_dim: int = 0,
                      text_dim: int = TEXT_DIM) -> "OrderedDict[str, tuple]":
    """kind = 'encoder' (GNNEncoder) or 'predictor' (GNNRetrosynthsizer)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["atom_encoder.weight"] = (N_ATOM_TYPES, H)
    s["virtualnode_embedding.weight"] = (1, H)
    if kind == "predictor":
        s["text_dropping.weight"] = (1, text_dim)
    for i in range(num_layer):
        p = f"convs.{i}."
        s[p + "eps"] = (1,)
        s[p + "mlp.0.weight"] = (4 * H, H)
        s[p + "mlp.0.bias"] = (4 * H,)
        s[p + "mlp.1.weight"] = (4 * H,)
        s[p + "mlp.1.bias"] = (4 * H,)
        s[p + "mlp.4.weight"] = (H, 4 * H)
        s[p + "mlp.4.bias"] = (H,)
        s[p + "bond_encoder.weight"] = (5, H)
        if kind == "encoder":
            s[f"norms.{i}.weight"] = (H,)
            s[f"norms.{i}.bias"] = (H,)
        else:
            s[f"adapters.{i}.1.weight"] = (3 * H, text_dim)
            s[f"adapters.{i}.1.bias"] = (3 * H,)
        if i < num_layer - 1:
            q = f"mlp_virtualnode_list.{i}."
            s[q + "0.weight"] = (4 * H, H)
            s[q + "0.bias"] = (4 * H,)
            s[q + "1.weight"] = (4 * H,)
            s[q + "1.bias"] = (4 * H,)
            s[q + "4.weight"] = (H, 4 * H)
            s[q + "4.bias"] = (H,)
    if kind == "predictor":
        s["decoder.0.weight"] = (4 * H, H)
        s["decoder.0.bias"] = (4 * H,)
        s["decoder.1.weight"] = (4 * H,)
        s["decoder.1.bias"] = (4 * H,)
        s["decoder.4.weight"] = (out_dim, 4 * H)
        s["decoder.4.bias"] = (out_dim,)
    return s


def proj_weight_shapes(H: int) -> "OrderedDict[str, tuple]":
    """ProjectionHead (reference graph_encoder/model.py:178-205)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["fc1.weight"] = (H, H)
    s["fc1.bias"] = (H,)
    s["norm1.weight"] = (H,)
    s["norm1.bias"] = (H,)
    s["fc2.weight"] = (H, H)
    s["fc2.bias"] = (H,)
    return s


def make_gin_weights(num_layer: int, H: int, kind: str, out_dim: int = 0, seed: int = 0, text_dim: int = TEXT_DIM):
    rs = np.random.RandomState(3000 + seed + (0 if kind == "encoder" else 500))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in gin_weight_shapes(num_layer, H, kind, out_dim, text_dim).items():
        if k.endswith("eps"):
            sd[k] = torch.from_numpy(rs.uniform(-0.2, 0.2, 1).astype(np.float32))
        else:
            sd[k] = torch.from_numpy(_fill(rs, k, shp))
    return sd


def make_proj_weights(H: int, seed: int = 0):
    rs = np.random.RandomState(4000 + seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in proj_weight_shapes(H).items():
        sd[k] = torch.from_numpy(_fill(rs, k, shp))
    return sd


def make_mol_graphs(n_graphs: int, seed: int = 0, min_atoms: int = 4, max_atoms: int = 32):
    """Synthetic molecule-like graphs in the integer encoding the reference feeds the GIN
    nets (modeling_llamole.py:720-760): ``x`` in [0,118), symmetric ``edge_index``
    (both directions listed, i->j then j->i), ``edge_attr`` in 1..4, ``batch`` sorted."""
    rs = np.random.RandomState(5000 + seed)
    xs, srcs, dsts, attrs, batch = [], [], [], [], []
    off = 0
    for g in range(n_graphs):
        n = int(rs.randint(min_atoms, max_atoms + 1))
        xs.append(rs.randint(0, N_ATOM_TYPES, n))
        bonds = set()
        for i in range(1, n):              # spanning tree, degree-capped
            for _ in range(8):
                j = int(rs.randint(0, i))
                if sum(1 for b in bonds if j in b) < 3:
                    break
            bonds.add((j, i))
        for _ in range(max(1, n // 10)):   # a few ring closures
            i, j = sorted(rs.choice(n, 2, replace=False).tolist()) if n > 2 else (0, 1)
            if i != j:
                bonds.add((i, j))
        for (i, j) in sorted(bonds):
            a = int(rs.randint(1, 5))
            srcs += [off + i, off + j]
            dsts += [off + j, off + i]
            attrs += [a, a]
        batch += [g] * n
        off += n
    x = torch.from_numpy(np.concatenate(xs).astype(np.int64))
    edge_index = torch.tensor([srcs, dsts], dtype=torch.int64)
    edge_attr = torch.tensor(attrs, dtype=torch.int64)
    return x, edge_index, edge_attr, torch.tensor(batch, dtype=torch.int64)


def make_cost_weights(seed: int = 0):
    """CostMLP(n_layers=1, fp_dim=2048, latent_dim=128) weights (reference
    graph_predictor/model.py:356-374: ``layers.0`` Linear(2048,128), ``layers.3`` Linear(128,1))."""
    rs = np.random.RandomState(7000 + seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in (("layers.0.weight", (128, 2048)), ("layers.0.bias", (128,)),
                   ("layers.3.weight", (1, 128)), ("layers.3.bias", (1,))):
        sd[k] = torch.from_numpy((0.05 * rs.standard_normal(shp)).astype(np.float32))
    return sd


def make_fingerprints(n: int, seed: int = 0):
    rs = np.random.RandomState(7500 + seed)
    return torch.from_numpy((rs.uniform(size=(n, 2048)) < 0.03).astype(np.float32))


def make_dit_train_batch(meta, B, seed, T):
    """Synthetic SFT batch for GraphDiT.forward: B molecules as PyG-style arrays (x = atom ids among the active atoms,
    symmetric edges with bond classes 1..4), properties with NaN / -200 slots, text rows, per-graph timesteps (t = 0, T and
    a repeated value included).  Returns x, edge_index, edge_attr, batch, props, text, t_int [B,1]."""
    import numpy as np
    import torch
    rs = np.random.RandomState(seed + 77)
    N = int(meta["max_node"])
    active = np.nonzero(np.asarray(meta["atom_type_dist"]) > 0)[0]
    xs, src, dst, att, bt, off = [], [], [], [], [], 0
    sizes = [N, max(2, N // 2), 3, 1] + [int(rs.randint(2, N + 1)) for _ in range(max(0, B - 4))]
    sizes = sizes[:B]
    for g, n in enumerate(sizes):
        xs.append(rs.choice(active, size=n))
        for i in range(1, n):                       # a random tree plus a few extra bonds
            j = int(rs.randint(0, i))
            a = int(rs.randint(1, 5))
            src += [off + i, off + j]
            dst += [off + j, off + i]
            att += [a, a]
        for _ in range(n // 4):
            i, j = rs.randint(0, n, size=2)
            if i != j and (off + i, off + j) not in set(zip(src, dst)):
                a = int(rs.randint(1, 5))
                src += [off + int(i), off + int(j)]
                dst += [off + int(j), off + int(i)]
                att += [a, a]
        bt += [g] * n
        off += n
    x = torch.from_numpy(np.concatenate(xs)).long()
    ei = torch.tensor([src, dst], dtype=torch.long).reshape(2, -1)
    ea = torch.tensor(att, dtype=torch.long)
    props, text, _ = make_dit_inputs(B, seed + 5, N)
    props[0, 1] = -200.0
    t = [0, T, 3, 3] + [int(rs.randint(1, T + 1)) for _ in range(max(0, B - 4))]
    return x, ei, ea, torch.tensor(bt, dtype=torch.long), props, text, torch.tensor(t[:B], dtype=torch.long).view(B, 1)


# --------------------------------------------------------------------------- whole-pipeline fixture (main.py eval on the GPU box)
EVAL_WORDS = ("design a molecule polymer drug with high low gas permeability co2 n2 o2 synthetic accessibility complexity that can "
              "you and the of to is has its for this be used analysis follow these procedures synthesize estimate remaining steps "
              "target consider following factors intermediate reagent availability side reactions stereochemistry challenges all "
              "readily available some commercial need mix multi step synthesis mostly require complex extensive given parameters "
              "current template reactants . , : ? 1 2 3 4 5").split()


def write_encoder_dir(path: str, num_layer: int = 3, H: int = 64, seed: int = 0) -> str:
    """GraphCLIP checkpoint directory (reference loader.py:322-363: config.json, model.pt, model_proj.pt)."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({"num_layer": num_layer, "hidden_size": H, "drop_ratio": 0.0}, f)
    torch.save(make_gin_weights(num_layer, H, "encoder", seed=seed), os.path.join(path, "model.pt"))
    torch.save(make_proj_weights(H, seed), os.path.join(path, "model_proj.pt"))
    return path


def write_predictor_dir(path: str, num_layer: int = 3, H: int = 64, out_dim: int = 512, seed: int = 0,
                        available=("R0", "R1", "R2", "R3", "A0", "A1", "A2", "B0", "B1", "B2", "B3", "B4", "B5")) -> str:
    """GraphPredictor checkpoint directory (reference loader.py:263-320: config.json, model.pt, cost_model.pt,
    label_to_template.csv.gz {rule_label, retro_templates}, available.csv.gz {smiles}); templates are named T<i>."""
    import pandas as pd
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({"num_layer": num_layer, "hidden_size": H, "drop_ratio": 0.0, "num_task": out_dim, "text_input_size": TEXT_DIM}, f)
    torch.save(make_gin_weights(num_layer, H, "predictor", out_dim, seed), os.path.join(path, "model.pt"))
    torch.save(make_cost_weights(seed), os.path.join(path, "cost_model.pt"))
    pd.DataFrame({"rule_label": list(range(out_dim)), "retro_templates": [f"T{i}" for i in range(out_dim)]}).to_csv(
        os.path.join(path, "label_to_template.csv.gz"), index=False, compression="gzip")
    pd.DataFrame({"smiles": list(available)}).to_csv(os.path.join(path, "available.csv.gz"), index=False, compression="gzip")
    return path


def write_llm_dir(path: str, special_tokens, name: str = "tiny", seed: int = 0, exact_vocab: bool = False) -> str:
    """A local `model_name_or_path`: tiny random-init HF causal LM (llamole_amd.e2e.LLM_CONFIGS[name]) saved with
    save_pretrained, plus a byte-level BPE tokenizer trained on a few sentences (the form AutoTokenizer resolves for a Qwen2 /
    Llama directory) with a chat template and an eos/pad token; the nine Llamole special tokens are NOT added yet (the driver
    adds them, like the reference's load_tokenizer, loader.py:88-138)."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast
    from . import e2e
    os.makedirs(path, exist_ok=True)
    tok = Tokenizer(models.BPE(unk_token=None))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=600, special_tokens=["<|endoftext|>", "<|user|>", "<|assistant|>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    corpus = [" ".join(EVAL_WORDS)] * 4 + ["CCO CC(=O)O c1ccccc1 N#C [H] * >> . T0 T1 R0.A1 To synthesize , follow these procedures :"]
    tok.train_from_iterator(corpus, trainer)
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|endoftext|>", pad_token="<|endoftext|>",
                                   additional_special_tokens=["<|user|>", "<|assistant|>"])
    fast.chat_template = ("{% for m in messages %}<|{{ m['role'] }}|> {{ m['content'] }} {% endfor %}"
                          "{% if add_generation_prompt %}<|assistant|> {% endif %}")
    fast.save_pretrained(path)
    # exact_vocab: the embedding matrices end where the base tokenizer ends, as a real base checkpoint's do -- adding the Llamole special
    # tokens then RESIZES them (reference patcher / adapter.py:224-233: the new rows are trained and saved with the adapter)
    llm = e2e.build_llm(name, "cpu", torch.bfloat16, seed=seed, **({"vocab_size": len(fast)} if exact_vocab else {}))
    assert exact_vocab or llm.config.vocab_size >= len(fast) + len(list(special_tokens))
    llm.save_pretrained(path, safe_serialization=True)
    return path


def write_connector_dir(path: str, llm_hidden: int, gin_hidden: int, seed: int = 0) -> str:
    """connector/{graph_to_lm_connector,lm_to_graph_decoder,lm_to_graph_predictor}.pt (reference modeling_llamole.py:246-275)."""
    os.makedirs(path, exist_ok=True)
    g = torch.Generator().manual_seed(8000 + seed)
    for name, (i, o) in (("graph_to_lm_connector", (gin_hidden, llm_hidden)), ("lm_to_graph_decoder", (llm_hidden, TEXT_DIM)),
                         ("lm_to_graph_predictor", (llm_hidden, TEXT_DIM))):
        torch.save({"0.weight": torch.randn(o, i, generator=g) * (1.0 / i) ** 0.5, "0.bias": torch.randn(o, generator=g) * 0.02},
                   os.path.join(path, name + ".pt"))
    return path


def write_lora_adapter_dir(path: str, llm, r: int = 4, alpha: int = 8, seed: int = 0,
                           targets=("q_proj", "v_proj", "down_proj"), modules_to_save=("embed_tokens", "lm_head")) -> str:
    """A LoRA adapter in peft's on-disk layout (adapter_config.json + adapter_model.safetensors with
    ``base_model.model.<module>.lora_A.weight`` / ``lora_B.weight`` keys) for every target Linear of ``llm``, plus full
    replacement weights for ``modules_to_save`` -- the resized embed_tokens / lm_head a reference adapter trained with
    resize_vocab carries (adapter.py:224-233); peft strips the adapter name, so they are stored as ``<module>.weight``."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    g = torch.Generator().manual_seed(9000 + seed)
    tensors = {}
    for name, mod in llm.named_modules():
        if name.split(".")[-1] in targets and isinstance(mod, torch.nn.Linear):
            tensors[f"base_model.model.{name}.lora_A.weight"] = torch.randn(r, mod.in_features, generator=g) * 0.05
            tensors[f"base_model.model.{name}.lora_B.weight"] = torch.randn(mod.out_features, r, generator=g) * 0.05
    for name, mod in llm.named_modules():
        if modules_to_save and name.split(".")[-1] in modules_to_save and hasattr(mod, "weight"):
            tensors[f"base_model.model.{name}.weight"] = (mod.weight.detach().float().cpu()
                                                           + 0.01 * torch.randn(mod.weight.shape, generator=g)).to(mod.weight.dtype)
    save_file(tensors, os.path.join(path, "adapter_model.safetensors"))
    with open(os.path.join(path, "adapter_config.json"), "w") as f:
        json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "target_modules": list(targets), "use_rslora": False,
                   "fan_in_fan_out": False, "bias": "none", "modules_to_save": list(modules_to_save) if modules_to_save else None}, f)
    return path


def write_molqa_dataset(dataset_dir: str, name: str = "molqa_synth", n: int = 5) -> str:
    """dataset_info.json + a MolQA-style json (instruction / input / property dict), reference data/*.json layout."""
    os.makedirs(dataset_dir, exist_ok=True)
    recs = []
    for i in range(n):
        recs.append({"instruction": "design a polymer with high co2 permeability and low synthetic complexity ?",
                     "input": "", "property": {"CO2": 10.0 + 7 * i, "N2": 1.0 + i, "SA": 3.0, "SC": 2.5}})
    with open(os.path.join(dataset_dir, "dataset_info.json"), "w") as f:
        json.dump({name: {"file_name": name + ".json"}}, f)
    with open(os.path.join(dataset_dir, name + ".json"), "w") as f:
        json.dump(recs, f)
    return name


def write_eval_fixture(root: str, special_tokens, with_adapter: bool = True, dit_shape=(128, 4, 4.0), gin_hidden: int = 64, n_prompts: int = 5,
                       batch_size: int = 2) -> str:
    """Everything `python main.py eval cfg.yaml` reads, synthetic and local: LLM + tokenizer, LoRA adapter + connectors, the
    three graph checkpoints, a dataset, and the YAML with the reference's keys (config/generate/qwen_material.yaml)."""
    import yaml
    from . import e2e
    llm_dir = write_llm_dir(os.path.join(root, "llm"), special_tokens)
    # dit_shape = (hidden_size, num_heads, mlp_ratio), gin_hidden: the checkpoints' own widths -- any the reference constructs
    cfg = make_dit_config(hidden_size=dit_shape[0], depth=2, num_heads=dit_shape[1], diffusion_steps=10, guide_scale=2.0, mlp_ratio=dit_shape[2])
    meta = make_data_meta(16, 0)
    write_dit_dir(os.path.join(root, "graph_decoder"), cfg, meta, make_dit_weights(cfg, 16, 0))
    write_encoder_dir(os.path.join(root, "graph_encoder"), H=gin_hidden)
    write_predictor_dir(os.path.join(root, "graph_predictor"), H=gin_hidden)
    hid = e2e.LLM_CONFIGS["tiny"]["hidden_size"]
    adapter = os.path.join(root, "adapter")
    if with_adapter:
        write_lora_adapter_dir(adapter, e2e.build_llm("tiny", "cpu", torch.bfloat16))
    write_connector_dir(os.path.join(adapter, "connector"), hid, gin_hidden)
    ds = write_molqa_dataset(os.path.join(root, "data"), n=n_prompts)
    y = {"model_name_or_path": llm_dir, "new_special_tokens": ",".join(special_tokens),
         "graph_decoder_path": os.path.join(root, "graph_decoder"), "graph_encoder_path": os.path.join(root, "graph_encoder"),
         "graph_predictor_path": os.path.join(root, "graph_predictor"), "adapter_name_or_path": adapter,
         "graph_lm_connector_path": os.path.join(adapter, "connector"), "stage": "mmsft", "do_train": False,
         "finetuning_type": "lora", "max_new_tokens": 16, "temperature": 0.6, "top_p": 0.9, "learned_query_size": 8,
         "dataset": ds, "dataset_dir": os.path.join(root, "data"), "template": "qwen", "cutoff_len": 32, "bf16": True,
         "pure_bf16": True, "per_device_eval_batch_size": batch_size, "output_dir": os.path.join(root, "out")}
    if not with_adapter:
        y.pop("adapter_name_or_path")
    path = os.path.join(root, "generate.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(y, f)
    return path


def write_molqa_train_dataset(dataset_dir: str, name: str = "molqa_train_synth", n: int = 6, out_dim: int = 512) -> str:
    """dataset_info.json entry + MolQA training records in the reference's text grammar (data/molqa_train_examples.json): a designed
    molecule after ``<design_start><design_end>``, one or two numbered retrosynthesis steps with ``<retro_start><retro_end>product>>reactants``,
    a property dict and one template label per step (``retro``; the last record's second step has none).  Molecule strings are short
    alphabetic names: whatever ``smiles_to_graph`` the caller installs must read them (tests use the ring-graph maker of tests/host_fakes.py)."""
    os.makedirs(dataset_dir, exist_ok=True)
    names = ["CCO", "CCN", "CCC", "COC", "CNC", "CCS", "CCF", "OCO"]
    step = ("This is step {i} in the retrosynthesis process. To synthesize <mol_start>{p}<mol_end>, follow these procedures: stir and heat. "
            "The applied reaction is: <retro_start><retro_end>{p}>>{a}.{b} with the template T{i}, which requires the reactants: {a} (available), {b} (available). ")
    recs = []
    for k in range(n):
        p, a, b, c = names[k % 8], names[(k + 1) % 8], names[(k + 2) % 8], names[(k + 3) % 8]
        out = f"To satisfy the requirements: a small polar scaffold. Therefore, the designed molecule is: <design_start><design_end><mol_start>{p}<mol_end>. "
        out += step.format(i=1, p=p, a=a, b=b)
        labels = [(37 * k + 5) % out_dim]
        if k % 2:
            out += step.format(i=2, p=a, a=b, b=c)
            labels.append(None if k == n - 1 else (91 * k + 11) % out_dim)
        recs.append({"instruction": f"design a molecule number {k} with low synthetic complexity ?", "input": "", "output": out,
                     "property": {"SA": 2.0 + 0.1 * k, "SC": 2.5, "BBBP": float(k % 2)}, "retro": labels})
    with open(os.path.join(dataset_dir, name + ".json"), "w") as f:
        json.dump(recs, f)
    info_path = os.path.join(dataset_dir, "dataset_info.json")
    info = {}
    if os.path.exists(info_path):
        with open(info_path) as f:
            info = json.load(f)
    info[name] = {"file_name": name + ".json"}
    with open(info_path, "w") as f:
        json.dump(info, f)
    return name


def write_train_fixture(root: str, special_tokens, **overrides) -> str:
    """Everything `python main.py train cfg.yaml` reads, synthetic and local, plus the YAML with the reference's training keys
    (config/train/mistral_lora.yaml)."""
    import yaml
    from . import e2e
    llm_dir = write_llm_dir(os.path.join(root, "llm"), special_tokens, exact_vocab=bool(overrides.pop("exact_vocab", False)))
    gin_hidden = int(overrides.pop("gin_hidden", 64))          # the GIN checkpoints' own width (any; 300 is the usual one of pretrained GINs)
    cfg = make_dit_config(hidden_size=128, depth=2, num_heads=4, diffusion_steps=10, guide_scale=2.0)
    write_dit_dir(os.path.join(root, "graph_decoder"), cfg, make_data_meta(16, 0), make_dit_weights(cfg, 16, 0))
    write_encoder_dir(os.path.join(root, "graph_encoder"), H=gin_hidden)
    write_predictor_dir(os.path.join(root, "graph_predictor"), H=gin_hidden)
    write_connector_dir(os.path.join(root, "connector0"), e2e.LLM_CONFIGS["tiny"]["hidden_size"], gin_hidden)
    ds = write_molqa_train_dataset(os.path.join(root, "data"))
    y = {"model_name_or_path": llm_dir, "new_special_tokens": ",".join(special_tokens),
         "graph_decoder_path": os.path.join(root, "graph_decoder"), "graph_encoder_path": os.path.join(root, "graph_encoder"),
         "graph_predictor_path": os.path.join(root, "graph_predictor"), "graph_lm_connector_path": None,
         "stage": "mmsft", "do_train": True, "finetuning_type": "lora", "lora_target": "all", "lora_rank": 4, "learned_query_size": 8,
         "dataset": ds, "dataset_dir": os.path.join(root, "data"), "template": "qwen", "cutoff_len": 256,
         "output_dir": os.path.join(root, "saves", "adapter"), "logging_steps": 1, "save_steps": 0, "overwrite_output_dir": True,
         "per_device_train_batch_size": 2, "gradient_accumulation_steps": 2, "learning_rate": 1.0e-3, "num_train_epochs": 2.0,
         "lr_scheduler_type": "cosine", "warmup_ratio": 0.1, "bf16": True, "pure_bf16": True,
         "loss_weight_retro": 1, "loss_weight_design": 1, "loss_weight_lm": 1}
    y.update(overrides)
    path = os.path.join(root, "train.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(y, f)
    return path
