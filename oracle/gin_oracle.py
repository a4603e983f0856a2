"""CPU oracle for the GIN encoder (GraphCLIP) and GIN predictor (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file; the shipped path never does.

Functional torch-CPU fp32 restatement of reference src/model/graph_encoder/model.py:87-205
and src/model/graph_predictor/model.py:231-423.  The scatter / pooling arithmetic lives in
the third-party dependency ``torch_geometric==2.6.1`` (reference requirements.txt:26), which
is absent from /root/reference; its published semantics are restated here:
  * ``MessagePassing(aggr="add").propagate(edge_index, x, edge_attr)`` with the default
    ``flow="source_to_target"``: message input ``x_j = x[edge_index[0]]``, summed into row
    ``edge_index[1]``;
  * ``global_add_pool`` / ``global_max_pool`` = segment sum / max over ``batch``.
Every edge list the reference builds is symmetric (modeling_llamole.py:749-751), so the
source/target convention cannot change a result.  Pinned against outputs of the reference
classes themselves (run with a PyG stub of exactly those semantics) in
``tests/golden/gin_*.npz`` -- PyG itself is not importable here, so parity at the PyG
boundary is pinned only to this restated semantics.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F


def _ln(x, w=None, b=None):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def _mlp4(sd, p, x):
    """Linear(H,4H) -> LayerNorm(4H, affine) -> GELU -> Linear(4H,H)  (graph_encoder/model.py:164)."""
    h = F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])
    h = F.gelu(_ln(h, sd[p + "1.weight"], sd[p + "1.bias"]))
    return F.linear(h, sd[p + "4.weight"], sd[p + "4.bias"])


def _segment_sum(h, batch, G):
    out = torch.zeros(G, h.shape[1], dtype=h.dtype)
    return out.index_add_(0, batch, h)


def _segment_max(h, batch, G):
    out = torch.full((G, h.shape[1]), float("-inf"), dtype=h.dtype)
    return out.scatter_reduce(0, batch[:, None].expand_as(h), h, reduce="amax", include_self=True)


def gin_conv(sd, p, h_in, edge_index, edge_attr):
    """graph_encoder/model.py:156-176 / graph_predictor/model.py:394-423."""
    msg = F.gelu(h_in[edge_index[0]] + sd[p + "bond_encoder.weight"][edge_attr])
    agg = torch.zeros_like(h_in).index_add_(0, edge_index[1], msg)
    return _mlp4(sd, p + "mlp.", (1 + sd[p + "eps"]) * h_in + agg)


def gin_trunk(sd: Dict[str, torch.Tensor], num_layer: int, x, edge_index, edge_attr, batch,
              c: Optional[torch.Tensor] = None, predictor: bool = False):
    """Encoder: graph_encoder/model.py:124-154.  Predictor: graph_predictor/model.py:306-353
    (LayerNorm without affine + modulate(shift, scale) and gated residual)."""
    G = int(batch[-1].item()) + 1
    vn = sd["virtualnode_embedding.weight"][torch.zeros(G, dtype=torch.long)]
    h = sd["atom_encoder.weight"][x]
    if predictor and c is None:
        c = sd["text_dropping.weight"].expand(G, -1)
    for l in range(num_layer):
        h_in = h + vn[batch]
        z = gin_conv(sd, f"convs.{l}.", h_in, edge_index, edge_attr)
        if predictor:
            m = F.linear(F.silu(c), sd[f"adapters.{l}.1.weight"], sd[f"adapters.{l}.1.bias"])
            shift, scale, gate = m.chunk(3, dim=1)
            z = _ln(z) * (1 + scale[batch]) + shift[batch]
        else:
            z = _ln(z, sd[f"norms.{l}.weight"], sd[f"norms.{l}.bias"])
        if l < num_layer - 1:
            z = F.gelu(z)
        h = (gate[batch] * z if predictor else z) + h_in
        if l < num_layer - 1:
            vn = vn + _mlp4(sd, f"mlp_virtualnode_list.{l}.", _segment_max(h_in, batch, G))
    return _segment_sum(h, batch, G)


def graphclip_forward(sd_enc, sd_proj, num_layer, x, edge_index, edge_attr, batch):
    """graph_encoder/model.py:37-41 + ProjectionHead :198-205."""
    g = gin_trunk(sd_enc, num_layer, x, edge_index, edge_attr, batch)
    h = F.linear(g, sd_proj["fc1.weight"], sd_proj["fc1.bias"])
    h = F.gelu(_ln(h, sd_proj["norm1.weight"], sd_proj["norm1.bias"]))
    h = F.linear(h, sd_proj["fc2.weight"], sd_proj["fc2.bias"])
    return h / h.norm(dim=-1, keepdim=True)


def predictor_forward(sd, num_layer, x, edge_index, edge_attr, batch, c):
    """graph_predictor/model.py:306-353 -> template logits [G,out_dim]."""
    g = gin_trunk(sd, num_layer, x, edge_index, edge_attr, batch, c, predictor=True)
    return _mlp4(sd, "decoder.", g)


def template_topk(logits, k):
    """graph_predictor/model.py:174-179: softmax(logits_main) only (the text-dropped pass is
    computed and discarded by the reference), then top-k."""
    return torch.topk(torch.softmax(logits, dim=1), k=k, dim=1)


def cost_mlp(sd, fps):
    """graph_predictor/model.py:356-391: softplus(Linear(128,1)(ReLU(Linear(2048,128)(fp))))."""
    h = F.relu(F.linear(fps, sd["layers.0.weight"], sd["layers.0.bias"]))
    h = F.linear(h, sd["layers.3.weight"], sd["layers.3.bias"])
    return torch.log(1 + torch.exp(h))
