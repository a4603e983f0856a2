"""Loader seam: drop-in for the graph loaders of reference ``src/model/loader.py:222-363``.

``load_graph_decoder / load_graph_predictor / load_graph_encoder(model_args, path, device)`` keep the reference
signatures and on-disk layouts (decoder ``config.yaml, data.meta.json, model.pt``; predictor ``config.json,
model.pt, cost_model.pt, label_to_template.csv.gz, available.csv.gz``; encoder ``config.json, model.pt,
model_proj.pt``).  Missing files are fetched from the same HuggingFace repos when the hub is reachable,
otherwise ``FileNotFoundError`` is raised (the reference behaves the same once its download fails).
"""
from __future__ import annotations

import json
import logging
from pathlib import Path

import torch

from .graph_decoder import GraphDiT
from .graph_encoder import GraphCLIP
from .graph_predictor import GraphPredictor

logger = logging.getLogger(__name__)

_REPOS = {"decoder": "liuganghuggingface/Llamole-Pretrained-GraphDiT",
          "predictor": "liuganghuggingface/Llamole-Pretrained-GNNPredictor",
          "encoder": "liuganghuggingface/Llamole-Pretrained-GraphEncoder"}


def _fetch(kind: str, filename: str, path: Path) -> Path:
    target = path / filename
    if target.exists():
        return target
    try:
        from huggingface_hub import hf_hub_download
        path.mkdir(parents=True, exist_ok=True)
        return Path(hf_hub_download(repo_id=_REPOS[kind], filename=filename, local_dir=str(path)))
    except Exception as e:
        raise FileNotFoundError(f"{target} not found and could not be downloaded from {_REPOS[kind]}: {e}") from e


def _finalize(model, model_args, device, what: str):
    if getattr(model_args, "disable_graph_model_gradient", True):
        model.disable_grads()
    model.to(device)
    compute_dtype = getattr(model_args, "compute_dtype", torch.float32)
    for p in model.parameters():     # reference loader.py:245-247: cast f32 params to the compute dtype
        if p.dtype == torch.float32 and compute_dtype != torch.float32:
            p.data = p.data.to(compute_dtype)
    n_all = sum(p.numel() for p in model.parameters())
    n_tr = sum(p.numel() for p in model.parameters() if p.requires_grad)
    logger.info("%s trainable params: %s || all params: %s", what, f"{n_tr:,}", f"{n_all:,}")
    return model


def load_graph_decoder(model_args, path: str, device):
    path = Path(path)
    config_path = _fetch("decoder", "config.yaml", path)
    for f in ("data.meta.json", "model.pt"):
        _fetch("decoder", f, path)
    model = GraphDiT(model_config_path=config_path, data_info_path=path / "data.meta.json",
                     model_dtype=getattr(model_args, "compute_dtype", torch.float32))
    model.init_model(path)
    return _finalize(model, model_args, device, "Graph DiT")


def load_graph_predictor(model_args, path: str, device):
    import pandas as pd
    path = Path(path)
    config_path = _fetch("predictor", "config.json", path)
    for f in ("model.pt", "cost_model.pt", "label_to_template.csv.gz", "available.csv.gz"):
        _fetch("predictor", f, path)
    with open(config_path, "r") as f:
        config = json.load(f)
    df = pd.read_csv(path / "label_to_template.csv.gz", compression="gzip")
    label_to_template = dict(zip(df["rule_label"], df["retro_templates"]))
    available = pd.read_csv(path / "available.csv.gz", compression="gzip")
    model = GraphPredictor(num_layer=config["num_layer"], hidden_size=config["hidden_size"],
                           drop_ratio=config["drop_ratio"], out_dim=config["num_task"], model_config=config,
                           label_to_template=label_to_template, available=available)
    model.init_model(path)
    model.init_neural_cost(path)
    model = _finalize(model, model_args, device, "Graph Predictor")
    if model.neural_cost is not None:
        model.neural_cost.to(device)
    return model


def load_graph_encoder(model_args, path: str, device):
    path = Path(path)
    config_path = _fetch("encoder", "config.json", path)
    for f in ("model.pt", "model_proj.pt"):
        _fetch("encoder", f, path)
    with open(config_path, "r") as f:
        config = json.load(f)
    model = GraphCLIP(graph_num_layer=config["num_layer"], graph_hidden_size=config["hidden_size"],
                      dropout=config["drop_ratio"], model_config=config)
    model.init_model(path, verbose=False)
    return _finalize(model, model_args, device, "Graph CLIP Encoder")
