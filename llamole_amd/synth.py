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


def make_gin_weights(num_layer: int, H: int, kind: str, out_dim: int = 0, seed: int = 0):
    rs = np.random.RandomState(3000 + seed + (0 if kind == "encoder" else 500))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in gin_weight_shapes(num_layer, H, kind, out_dim).items():
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
