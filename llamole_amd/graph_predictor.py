"""GraphPredictor (conditional GIN template classifier + CostMLP) on MI355X: drop-in for reference
``src/model/graph_predictor/model.py:GraphPredictor``.

Same constructor / attributes / methods: ``GraphPredictor(num_layer, hidden_size,
drop_ratio, out_dim, model_config, label_to_template, available)``, ``.text_input_size``, ``.available``,
``.label_to_template``, ``forward(x, edge_index, edge_attr, batch, c)``, ``sample_templates(product_graph, c,
product_smiles, topk)``, ``estimate_cost(smiles)``, ``init_model / init_neural_cost / save_pretrained``.
GIN trunk, decoder GEMV, softmax + top-k and the CostMLP run in the HIP library; template application
(rdchiral) and Morgan fingerprints (rdkit) are host chemistry exactly as in the reference and are imported
lazily -- their absence raises ImportError at the call, never a silent fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from collections import defaultdict
from typing import Callable, List, Optional

import numpy as np
import torch

from . import _lib
from .graph_encoder import check_graph_errors, csr_for_engine, graph_csr, _GinModule
from .synth import gin_weight_shapes
from .weights import WeightBag


def _default_template_runner() -> Callable[[str, str], List[str]]:
    try:
        from rdchiral.main import rdchiralRunText
    except ImportError as e:  # pragma: no cover - depends on the environment
        raise ImportError("GraphPredictor.sample_templates needs `rdchiral` (reference requirements.txt:21) to apply "
                          "retro templates; pass template_runner=... to inject one") from e
    return rdchiralRunText


def merge_template_outcomes(topk_probs, templates, product_smiles, run_template):
    """Host tail of sample_templates (reference graph_predictor/model.py:190-228): apply each template,
    split its probability evenly over its outcomes, canonicalise multi-reactant outcomes by sorting the
    dot-separated parts, sum scores per reactant set (first template kept), sort descending, renormalise."""
    by_reactant = defaultdict(list)
    for prob, template in zip(topk_probs, templates):
        try:
            outcomes = run_template(template, product_smiles)
            if len(outcomes) == 0:
                continue
            outcomes = sorted(outcomes)
            share = float(prob) / len(outcomes)
            for reactant in outcomes:
                key = ".".join(sorted(reactant.strip().split("."))) if "." in reactant else reactant
                by_reactant[key].append((share, template))
        except Exception:
            pass
    if not by_reactant:
        return [], [], []
    merged = [(r, sum(s for s, _ in lst), lst[0][1]) for r, lst in by_reactant.items()]
    merged.sort(key=lambda it: it[1], reverse=True)   # stable: ties keep first-seen order, like sorted() in the reference
    reactants = [m[0] for m in merged]
    scores = [m[1] for m in merged]
    tmpls = [m[2] for m in merged]
    total = sum(scores)
    return reactants, [s / total for s in scores], tmpls


class _PredictorWithGrad(torch.autograd.Function):
    """logits = predictor(graphs, c) with d logits -> d c computed by the HIP engine (weights frozen)."""

    @staticmethod
    def forward(ctx, c, module, x, edge_index, edge_attr, batch):
        c32 = c.detach().to(device=module._device(), dtype=torch.float32).contiguous()
        out, saved = module._forward_train(x, edge_index, edge_attr, batch, c32)
        ctx.module, ctx.saved, ctx.c32, ctx.c_dtype = module, saved, c32, c.dtype
        return out.to(next(module.predictor.parameters()).dtype)

    @staticmethod
    def backward(ctx, dlogits):
        dc = ctx.module._backward_c(ctx.saved, ctx.c32, dlogits)
        return dc.to(ctx.c_dtype), None, None, None, None, None


class GraphPredictor(_GinModule):
    def __init__(self, num_layer, hidden_size, drop_ratio, out_dim, model_config, label_to_template, available=None):
        super().__init__()
        if num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")
        if not 1 <= int(hidden_size) <= 2048:      # any width runs (zero-padded to a multiple of 64 inside the engine, csrc/gin.hip: GinDims) up to:
            raise ValueError(f"hidden_size={hidden_size}: the MI355X GIN engine handles hidden_size <= 2048")
        self.model_config = model_config
        self.text_input_size = model_config.get("text_input_size", 768)
        self.available = available
        self.text_drop = drop_ratio
        self.num_layer, self.hidden_size, self.out_dim = num_layer, hidden_size, out_dim
        try:
            import pandas as pd
            is_df = isinstance(label_to_template, pd.DataFrame)
        except ImportError:  # pragma: no cover
            is_df = False
        if is_df:
            self.label_to_template = dict(zip(label_to_template["rule_label"], label_to_template["retro_templates"]))
        else:
            self.label_to_template = label_to_template
        self.predictor = WeightBag(gin_weight_shapes(num_layer, hidden_size, "predictor", out_dim, self.text_input_size))
        self.neural_cost = None
        self._cost_arena = None
        self._handle = None
        self.template_runner: Optional[Callable[[str, str], List[str]]] = None

    # ------------------------------------------------------------------ engine plumbing
    def _bags(self):
        return (self.predictor,)

    def _gin_cfg(self, code):
        return _lib.LLGinConfig(self.num_layer, self.hidden_size, 1, self.out_dim, self.text_input_size, code)

    def _named_for_arena(self):
        return self.predictor.state_dict().items()

    # ------------------------------------------------------------------ reference surface
    def forward(self, x, edge_index, edge_attr, batch, c):
        """Template logits [G, out_dim].  With autograd enabled and ``c.requires_grad`` (SFT: the retro cross-entropy trains
        the LLM through ``c``, reference modeling_llamole.py:385-419; the predictor itself is frozen) the result carries a
        grad_fn whose backward is the HIP reverse sweep ``ll_gin_backward_c``."""
        if c is not None and torch.is_grad_enabled() and c.requires_grad:
            return _PredictorWithGrad.apply(c, self, x, edge_index, edge_attr, batch)
        with torch.no_grad():
            out = self._run(x, edge_index, edge_attr, batch, c, self.out_dim)
            return out.to(next(self.predictor.parameters()).dtype)

    def _forward_train(self, x, edge_index, edge_attr, batch, c32):
        self._ensure_engine()
        dev = self._device()
        x, edge_index, edge_attr, batch = x.to(dev), edge_index.to(dev), edge_attr.to(dev), batch.to(dev)
        ng = getattr(batch, "_ll_num_graphs", None) or int(c32.shape[0])     # one condition row per graph
        xs, rowptr, src, attr, b, gptr, n, ne, G = csr_for_engine(x, edge_index, edge_attr, batch, ng)
        if c32.shape[0] != G:
            raise ValueError(f"condition rows {c32.shape[0]} != number of graphs {G}")
        # the reverse sweep walks OUT-edges: CSR keyed by source = the CSR of the flipped edge list
        _, rowptr_s, dst_s, attr_s, _, _, _, _, _ = csr_for_engine(x, edge_index.flip(0), edge_attr, batch, ng)
        out = torch.empty(G, self.out_dim, device=dev, dtype=torch.float32)
        _lib.check(_lib.load().ll_gin_forward_train(self._handle, _lib.dptr(xs), _lib.dptr(rowptr), _lib.dptr(src), _lib.dptr(attr),
                                                    _lib.dptr(b), _lib.dptr(gptr), n, ne, G, _lib.dptr(c32), _lib.dptr(out),
                                                    _lib.current_stream_ptr()), "ll_gin_forward_train")
        return out, (rowptr_s, dst_s, attr_s, b, gptr, n, ne, G)       # stream-ordered: no host synchronisation needed

    def _backward_c(self, saved, c32, dlogits):
        rowptr_s, dst_s, attr_s, b, gptr, n, ne, G = saved
        check_graph_errors(wait=True)      # the forward's batch conversion is long done: an unsorted batch / bad edge raises here
        dlogits = dlogits.to(torch.float32).contiguous()
        dc = torch.empty_like(c32)
        _lib.check(_lib.load().ll_gin_backward_c(self._handle, _lib.dptr(rowptr_s), _lib.dptr(dst_s), _lib.dptr(attr_s), _lib.dptr(b),
                                                 _lib.dptr(gptr), n, ne, G, _lib.dptr(c32), _lib.dptr(dlogits), _lib.dptr(dc),
                                                 _lib.current_stream_ptr()), "ll_gin_backward_c")
        return dc

    @torch.no_grad()
    def topk_templates(self, x, edge_index, edge_attr, batch, c, topk: int, num_graphs=None):
        """Device part of sample_templates: logits -> softmax -> top-k (model.py:174-179).
        The reference also evaluates the text-dropped predictor and discards it (:175-177); that dead
        pass is not reproduced."""
        logits = self._run(x, edge_index, edge_attr, batch, c, self.out_dim, num_graphs=num_graphs)
        G = logits.shape[0]
        k = min(int(topk), self.out_dim)
        if k > 64:
            raise ValueError("top-k above 64 is not supported by ll_softmax_topk")
        probs = torch.empty(G, k, device=logits.device, dtype=torch.float32)
        idx = torch.empty(G, k, device=logits.device, dtype=torch.int32)
        _lib.check(_lib.load().ll_softmax_topk(_lib.dptr(logits), G, self.out_dim, k, _lib.dptr(probs), _lib.dptr(idx),
                                               _lib.current_stream_ptr()), "ll_softmax_topk")
        return probs, idx        # on the current stream; the caller's .cpu() is the only synchronisation

    @torch.no_grad()
    def sample_templates(self, product_graph, c, product_smiles, topk=10):
        x, edge_index, edge_attr = product_graph.x, product_graph.edge_index, product_graph.edge_attr
        batch = torch.zeros(x.size(0), dtype=torch.long, device=x.device)
        probs, idx = self.topk_templates(x, edge_index, edge_attr, batch, c, topk, num_graphs=1)
        topk_probs = probs.float().cpu().numpy()[0]
        topk_indices = idx.cpu().numpy()[0]
        templates = [self.label_to_template[int(i)] for i in topk_indices]
        run = self.template_runner or _default_template_runner()
        return merge_template_outcomes(topk_probs, templates, product_smiles, run)

    @torch.no_grad()
    def sample_templates_batch(self, product_graphs, c, product_smiles_list, topk=10):
        """``sample_templates`` for several products with ONE GIN forward + one top-k launch (batched
        expansions of concurrent A* searches); the host tail (template application, merge) runs per product.
        product_graphs: list of GraphData; c [G, text_input_size]; returns a list of (reactants, scores, templates)."""
        probs, idx = self.topk_templates_batch(product_graphs, c, topk)
        return self.merge_topk(probs.float().cpu().numpy(), idx.cpu().numpy(), product_smiles_list)

    @torch.no_grad()
    def topk_templates_batch(self, product_graphs, c, topk=10):
        """Device half of ``sample_templates_batch``: one GIN forward + one top-k launch -> (probs [G, k] f32, idx [G, k] int32) on the
        device -- the fixed-size per-expansion record ranks exchange when a lock-step round is sharded (distributed.all_gather_topk)."""
        from .graph_data import GraphBatch
        gb = GraphBatch.from_data_list(list(product_graphs))
        return self.topk_templates(gb.x, gb.edge_index, gb.edge_attr, gb.batch, c, topk)

    def merge_topk(self, probs, idx, product_smiles_list):
        """Host half: template application and the reference's merge (model.py:190-228) per product; probs / idx numpy [G, k]."""
        run = self.template_runner or _default_template_runner()
        out = []
        for g, smiles in enumerate(product_smiles_list):
            templates = [self.label_to_template[int(i)] for i in idx[g]]
            out.append(merge_template_outcomes(probs[g], templates, smiles, run))
        return out

    # cost model ------------------------------------------------------------------------------
    def init_neural_cost(self, model_path, verbose=False):
        model_file = os.path.join(model_path, "cost_model.pt")
        if not os.path.exists(model_file):
            raise FileNotFoundError(f"Model file not found: {model_file}")
        self.neural_cost = WeightBag({"layers.0.weight": (128, 2048), "layers.0.bias": (128,),
                                      "layers.3.weight": (1, 128), "layers.3.bias": (1,)})
        self.neural_cost.load_state_dict(torch.load(model_file, map_location="cpu", weights_only=True))
        self.neural_cost.to(next(self.predictor.parameters()).device)
        self._cost_arena = None

    @staticmethod
    def smiles_to_fp(smiles: str, fp_dim: int = 2048) -> np.ndarray:
        """Morgan radius-2 bit fingerprint (CostMLP.smiles_to_fp, model.py:374-383); host chemistry."""
        try:
            from rdkit import Chem
            from rdkit.Chem import AllChem
        except ImportError as e:  # pragma: no cover
            raise ImportError("estimate_cost needs `rdkit` for Morgan fingerprints (reference requirements.txt:22)") from e
        mol = Chem.MolFromSmiles(smiles)
        if mol is None:
            raise ValueError(f"Invalid SMILES string: {smiles}")
        fp = AllChem.GetMorganFingerprintAsBitVect(mol, 2, nBits=fp_dim)
        arr = np.zeros(fp.GetNumBits(), dtype=bool)
        arr[list(fp.GetOnBits())] = 1
        return arr

    @torch.no_grad()
    def cost_from_fingerprints(self, fps: torch.Tensor) -> torch.Tensor:
        """CostMLP.forward on the device (model.py:385-391); fps [n,2048] of 0/1."""
        if self.neural_cost is None:
            raise ValueError("Cost model is not initialized.")
        dev = self._device()
        if self._cost_arena is None or self._cost_fp != self.neural_cost.fingerprint():
            sd = self.neural_cost.state_dict()
            self._cost_arena = torch.cat([sd[k].detach().float().reshape(-1).to(dev) for k in
                                          ("layers.0.weight", "layers.0.bias", "layers.3.weight", "layers.3.bias")]).contiguous()
            self._cost_fp = self.neural_cost.fingerprint()
        fps = fps.to(device=dev, dtype=torch.float32).contiguous()
        out = torch.empty(fps.shape[0], device=dev, dtype=torch.float32)
        _lib.check(_lib.load().ll_cost_mlp(_lib.dptr(self._cost_arena), _lib.dptr(fps), fps.shape[0], _lib.dptr(out),
                                           _lib.current_stream_ptr()), "ll_cost_mlp")
        torch.cuda.current_stream().synchronize()
        return out

    def estimate_cost(self, smiles):
        if self.neural_cost is None:
            raise ValueError("Cost model is not initialized.")
        fp = torch.from_numpy(self.smiles_to_fp(smiles).astype(np.float32)).view(1, -1)
        return float(self.cost_from_fingerprints(fp)[0].item())

    # persistence -------------------------------------------------------------------------------
    def init_model(self, model_path, verbose=False):
        model_file = os.path.join(model_path, "model.pt")
        if not os.path.exists(model_file):
            raise FileNotFoundError(f"Model file not found: {model_file}")
        self.predictor.load_state_dict(torch.load(model_file, map_location="cpu", weights_only=True))

    def save_pretrained(self, output_dir):
        import pandas as pd
        os.makedirs(output_dir, exist_ok=True)
        torch.save(self.predictor.state_dict(), os.path.join(output_dir, "model.pt"))
        if self.neural_cost is not None:
            torch.save(self.neural_cost.state_dict(), os.path.join(output_dir, "cost_model.pt"))
        with open(os.path.join(output_dir, "model_config.json"), "w") as f:
            json.dump(self.model_config, f, indent=2)
        pd.DataFrame(list(self.label_to_template.items()), columns=["rule_label", "retro_templates"]).to_csv(
            os.path.join(output_dir, "label_to_template.csv.gz"), index=False, compression="gzip")
        if self.available is not None:
            if isinstance(self.available, list):
                df = pd.DataFrame(self.available, columns=["smiles"])
            elif isinstance(self.available, pd.DataFrame):
                df = self.available
            else:
                raise ValueError("available must be either a list of SMILES strings or a pandas DataFrame")
            df.to_csv(os.path.join(output_dir, "available.csv.gz"), index=False, compression="gzip")

    def disable_grads(self):
        for p in self.predictor.parameters():
            p.requires_grad = False
