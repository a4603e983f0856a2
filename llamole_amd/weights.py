"""Weight containers that keep the reference's state-dict key names.

The reference modules are ordinary ``nn.Module`` trees; checkpoints (``model.pt``) are their
``state_dict()``s (reference diffusion_model.py:116-121, graph_encoder/model.py:72-81,
graph_predictor/model.py:138-146).  The MI355X engines do not use nn.Module arithmetic at all --
they read one flat f32 arena -- but the drop-in classes must still ``load_state_dict`` those
checkpoints, expose ``parameters()`` (the reference loader casts them, loader.py:245-247) and
``save_pretrained`` the same keys.  ``WeightBag`` is that: a parameter-only module tree built from a
``{dotted.key: shape}`` table.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Tuple

import torch
import torch.nn as nn

from . import _lib


class WeightBag(nn.Module):
    """Parameter-only module whose ``state_dict()`` keys equal the given dotted names."""

    def __init__(self, shapes: Dict[str, Tuple[int, ...]]):
        super().__init__()
        for key, shape in shapes.items():
            parts = key.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(*shape), requires_grad=False))
        self._version_counter = 0

    def forward(self, *a, **k):  # pragma: no cover - arithmetic lives in the HIP engine
        raise RuntimeError("WeightBag holds parameters only; the forward pass runs in the HIP engine")

    def fingerprint(self) -> tuple:
        """Cheap change detector: (data_ptr, _version, dtype, device) of every parameter."""
        return tuple((p.data_ptr(), p._version, p.dtype, str(p.device)) for p in self.parameters())


def pack_arena(kind: str, cfg, named: Iterable[Tuple[str, torch.Tensor]], device) -> torch.Tensor:
    """Copy parameters into the flat f32 device arena whose layout the C library defines
    (``ll_{dit,gin}_param_info``).  Missing or mis-sized parameters raise."""
    lib = _lib.load()
    table = _lib.param_table(kind, cfg)
    total = getattr(lib, f"ll_{kind}_arena_elems")(C.byref(cfg))
    arena = torch.zeros(int(total), dtype=torch.float32, device=device)
    named = dict(named)
    for name, numel, off in table:
        if name not in named:
            raise KeyError(f"parameter '{name}' required by the HIP engine is missing from the state dict")
        t = named[name]
        if t.numel() != numel:
            raise ValueError(f"parameter '{name}': expected {numel} elements, got {tuple(t.shape)}")
        arena[off:off + numel] = t.detach().to(device=device, dtype=torch.float32).reshape(-1)
    return arena


def engine_dtype(torch_dtype) -> int:
    if torch_dtype == torch.float32:
        return _lib.LL_F32
    if torch_dtype in (torch.bfloat16, torch.float16):
        return _lib.LL_BF16   # 16-bit checkpoints run on the bf16 MFMA path
    raise ValueError(f"unsupported model dtype {torch_dtype}")
