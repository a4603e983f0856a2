"""GraphCLIP (GIN encoder + projection head) on MI355X: drop-in for reference
``src/model/graph_encoder/model.py:GraphCLIP``.

``GraphCLIP(graph_num_layer, graph_hidden_size, dropout, model_config)``; ``.hidden_size``;
``forward(x, edge_index, edge_attr, batch) -> [G, H]`` unit-norm embeddings; ``init_model(path)``
loads ``model.pt`` (GNNEncoder keys) and ``model_proj.pt`` (ProjectionHead keys);
``save_pretrained`` writes the same three files.  The forward runs in the HIP engine (ll_gin_forward).
"""
from __future__ import annotations

import ctypes as C
import json
import os

import torch
import torch.nn as nn

from . import _lib
from .synth import gin_weight_shapes, proj_weight_shapes
from .weights import WeightBag, engine_dtype, pack_arena


def graph_csr(x, edge_index, edge_attr, batch):
    """PyG-style (edge_index [2,E], edge_attr [E], batch [n]) -> the int32 CSR-by-destination arrays the C ABI takes.
    Stable sort by destination keeps the reference's per-destination summation order."""
    dev = x.device
    n = x.shape[0]
    dst = edge_index[1].long()
    order = torch.argsort(dst, stable=True)
    src = edge_index[0].long()[order].to(torch.int32).contiguous()
    attr = edge_attr.long()[order].to(torch.int32).contiguous()
    deg = torch.bincount(dst, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
    b = batch.long()
    G = int(b[-1].item()) + 1
    if bool((b[1:] < b[:-1]).any()):
        raise ValueError("`batch` must be sorted (PyG Batch convention)")
    gptr = torch.zeros(G + 1, dtype=torch.int32, device=dev)
    gptr[1:] = torch.cumsum(torch.bincount(b, minlength=G), 0).to(torch.int32)
    return (x.to(torch.int32).contiguous(), rowptr.contiguous(), src, attr, b.to(torch.int32).contiguous(),
            gptr.contiguous(), n, int(src.numel()), G)


CSR_KERNEL_MAX_NODES, CSR_KERNEL_MAX_EDGES = 32768, 262144     # one workgroup builds the CSR; beyond this the ATen route


_CSR_ERRORS = {1: "`batch` must be sorted (PyG Batch convention) with ids in [0, num_graphs)",
               2: "edge_index refers to a node outside the batch",
               3: "atom type outside [0, 118) or bond type outside [0, 5)"}
_pending_csr_flags = []       # slots of the pinned flag ring whose conversion kernel may not have finished yet
_flag_origin = {}             # slot -> "file:line (function)" of the call that converted the batch (named in the error message)
_flag_ring = None             # pinned int32[256]: the conversion kernel's LAST instruction stores its error flag straight into host memory
_flag_next = 0


def _flag_slot():
    global _flag_ring, _flag_next
    if _flag_ring is None:
        _flag_ring = torch.zeros(256, dtype=torch.int32).pin_memory()
    if len(_pending_csr_flags) >= 128:
        check_graph_errors(wait=True)
    slot = _flag_next
    _flag_next = (_flag_next + 1) % 256
    _flag_ring[slot] = -1         # "not run yet"; the kernel overwrites it with 0 (ok) or an error code when it is done
    _flag_origin[slot] = _caller()
    return slot


def _caller() -> str:
    """First frame outside the graph wrappers: the call that handed over the batch (a late error must name IT, not the call that finds the flag)."""
    import sys
    f = sys._getframe(2)
    here = ("graph_encoder.py", "graph_predictor.py")
    while f is not None and f.f_code.co_filename.endswith(here):
        f = f.f_back
    return "?" if f is None else f"{f.f_code.co_filename.rsplit('/', 1)[-1]}:{f.f_lineno} ({f.f_code.co_name})"


def check_graph_errors(wait: bool = False):
    """Raise ValueError if an earlier ``graph_csr_device`` call saw a malformed batch.  The conversion kernel clamps every id it
    writes (so the GIN kernels never index out of bounds) and, as its last instruction, stores a flag into pinned host memory;
    this looks at the flags that have ARRIVED -- no copy, no event and no synchronisation on the hot path (an event record between
    the conversion and the forward cost ~6 us of idle GPU per call).  ``wait=True`` (end of an eval / generation entry point, before
    outputs are read on the host, reverse sweeps): flags still outstanding after the poll are waited for -- every device that has one
    pending is synchronised, and only then; a conversion that has long finished never drains a queue.  The message names the call
    that converted the offending batch."""
    def poll():
        keep, bad = [], None
        for slot in _pending_csr_flags:
            v = int(_flag_ring[slot])
            if v < 0:
                keep.append(slot)
            elif v > 0 and (bad is None or v > bad[0]):
                bad = (v, _flag_origin.get(slot, "?"))
        _pending_csr_flags[:] = keep
        return bad
    bad = poll()
    if bad is None and wait and _pending_csr_flags:
        for d in range(torch.cuda.device_count()):
            torch.cuda.synchronize(d)
        bad = poll()
    if bad is not None:
        code, origin = bad
        raise ValueError(_CSR_ERRORS.get(code, f"malformed graph batch (code {code})") + f" [batch converted at {origin}]")


def graph_csr_device(x, edge_index, edge_attr, batch, num_graphs=None):
    """``graph_csr`` in ONE launch on the HIP device (``ll_graph_csr``).  With ``num_graphs`` given (``GraphBatch`` tags its
    ``batch`` tensor with it, single-graph callers pass 1) nothing synchronises with the host: the kernel's error flag (unsorted
    ``batch``, edge / atom / bond id out of range -- all clamped, so whatever runs on the arrays stays in bounds) is written by the
    kernel into pinned host memory and raised by the next call that finds it (``check_graph_errors``).  Otherwise the graph count is
    read back from ``batch[-1]`` like the ATen route does and the flag is checked right away."""
    check_graph_errors()
    dev = x.device
    n, ne = int(x.shape[0]), int(edge_index.shape[1])
    known = num_graphs is not None
    G = int(num_graphs) if known else int(batch[-1].item()) + 1
    i32 = dict(dtype=torch.int32, device=dev)
    xs, rowptr, b32, gptr = torch.empty(n, **i32), torch.empty(n + 1, **i32), torch.empty(n, **i32), torch.empty(G + 1, **i32)
    src, attr = torch.empty(ne, **i32), torch.empty(ne, **i32)
    scratch = torch.empty(n, **i32)
    x64, ei64, ea64, b64 = (t.long().contiguous() for t in (x, edge_index, edge_attr, batch))
    if known:
        slot = _flag_slot()
        err_ptr = C.c_void_p(_flag_ring.data_ptr() + 4 * slot)      # pinned host memory is device-accessible at the same address
    else:
        err = torch.empty(1, **i32)
        err_ptr = _lib.dptr(err)
    _lib.check(_lib.load().ll_graph_csr(_lib.dptr(x64), _lib.dptr(ei64) if ne else None, _lib.dptr(ea64) if ne else None, _lib.dptr(b64),
                                        n, ne, G, _lib.dptr(xs), _lib.dptr(rowptr), _lib.dptr(src) if ne else None,
                                        _lib.dptr(attr) if ne else None, _lib.dptr(b32), _lib.dptr(gptr), _lib.dptr(scratch),
                                        err_ptr, _lib.current_stream_ptr()), "ll_graph_csr")
    if not known:
        code = int(err.item())
        if code:
            raise ValueError(_CSR_ERRORS.get(code, f"malformed graph batch (code {code})"))
    else:
        _pending_csr_flags.append(slot)
    return xs, rowptr, src, attr, b32, gptr, n, ne, G


def csr_for_engine(x, edge_index, edge_attr, batch, num_graphs=None):
    """CSR arrays for ll_gin_forward: the one-launch kernel on the device, the ATen route for very large batches / CPU tensors."""
    if num_graphs is None:
        num_graphs = getattr(batch, "_ll_num_graphs", None)
    if x.is_cuda and x.shape[0] <= CSR_KERNEL_MAX_NODES and edge_index.shape[1] <= CSR_KERNEL_MAX_EDGES:
        return graph_csr_device(x, edge_index, edge_attr, batch, num_graphs)
    return graph_csr(x, edge_index, edge_attr, batch)


class _GinModule(nn.Module):
    """Shared engine plumbing of the encoder and the predictor."""

    _kind = 0

    def _gin_cfg(self, code):
        raise NotImplementedError

    def _named_for_arena(self):
        raise NotImplementedError

    def _bags(self):
        raise NotImplementedError

    def _device(self):
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} runs on the HIP device only: call .to('cuda') first (no CPU path)")
        return p.device

    def _ensure_engine(self):
        dev = self._device()
        fp = tuple(b.fingerprint() for b in self._bags())
        if getattr(self, "_handle", None) is not None and fp == self._fingerprint:
            return
        self._release()
        lib = _lib.load()
        p0 = next(self.parameters())
        code = engine_dtype(p0.dtype)
        cfg = self._gin_cfg(code)
        with torch.cuda.device(dev):
            self._arena = pack_arena("gin", cfg, self._named_for_arena(), dev)
            torch.cuda.synchronize(dev)
            h = C.c_void_p()
            _lib.check(lib.ll_gin_create(C.byref(cfg), _lib.dptr(self._arena), C.byref(h)), "ll_gin_create")
        self._handle, self._fingerprint = h, fp

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.load().ll_gin_destroy(self._handle)
        self._handle = None
        self._arena = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _run(self, x, edge_index, edge_attr, batch, c, out_cols, want_pooled=False, num_graphs=None):
        self._ensure_engine()
        dev = self._device()
        if num_graphs is None:
            num_graphs = getattr(batch, "_ll_num_graphs", None)      # GraphBatch tags its batch vector: no read-back of batch[-1]
        xs, rowptr, src, attr, b, gptr, n, ne, G = csr_for_engine(x.to(dev), edge_index.to(dev), edge_attr.to(dev), batch.to(dev), num_graphs)
        out = torch.empty(G, out_cols, device=dev, dtype=torch.float32)
        pooled = torch.empty(G, self.hidden_size, device=dev, dtype=torch.float32) if want_pooled else None
        if c is not None:
            c = c.detach().to(device=dev, dtype=torch.float32).contiguous()
            if c.shape[0] != G:
                raise ValueError(f"condition rows {c.shape[0]} != number of graphs {G}")
        _lib.check(_lib.load().ll_gin_forward(self._handle, _lib.dptr(xs), _lib.dptr(rowptr), _lib.dptr(src), _lib.dptr(attr),
                                              _lib.dptr(b), _lib.dptr(gptr), n, ne, G, _lib.dptr(c), _lib.dptr(out),
                                              _lib.dptr(pooled), _lib.current_stream_ptr()), "ll_gin_forward")
        # no host synchronisation: the temporaries above come from torch's stream-ordered caching allocator and every launch of
        # the engine went to torch's current stream, so their memory cannot be handed out again before those launches have run
        return (out, pooled) if want_pooled else out

    def disable_grads(self):
        for p in self.parameters():
            p.requires_grad = False


class GraphCLIP(_GinModule):
    def __init__(self, graph_num_layer, graph_hidden_size, dropout, model_config):
        super().__init__()
        if graph_num_layer < 2:
            raise ValueError("Number of GNN layers must be greater than 1.")
        if not 1 <= int(graph_hidden_size) <= 2048:      # any width runs (zero-padded to a multiple of 64 inside the engine, csrc/gin.hip: GinDims) up to:
            raise ValueError(f"hidden_size={graph_hidden_size}: the MI355X GIN engine handles hidden_size <= 2048")
        self.model_config = model_config
        self.hidden_size = graph_hidden_size
        self.num_layer = graph_num_layer
        self.molecule_encoder = WeightBag(gin_weight_shapes(graph_num_layer, graph_hidden_size, "encoder"))
        self.molecule_projection = WeightBag(proj_weight_shapes(graph_hidden_size))
        self._handle = None

    def _bags(self):
        return (self.molecule_encoder, self.molecule_projection)

    def _gin_cfg(self, code):
        return _lib.LLGinConfig(self.num_layer, self.hidden_size, 0, 0, 768, code)

    def _named_for_arena(self):
        d = dict(self.molecule_encoder.state_dict())
        d.update({"proj." + k: v for k, v in self.molecule_projection.state_dict().items()})
        return d.items()

    @torch.no_grad()
    def forward(self, x, edge_index, edge_attr, batch):
        out = self._run(x, edge_index, edge_attr, batch, None, self.hidden_size)
        return out.to(next(self.parameters()).dtype)

    @torch.no_grad()
    def pooled(self, x, edge_index, edge_attr, batch):
        """Add-pooled node states before the projection head (GNNEncoder.forward output)."""
        return self._run(x, edge_index, edge_attr, batch, None, self.hidden_size, want_pooled=True)[1]

    def init_model(self, model_path, verbose=True):
        molecule_path = os.path.join(model_path, "model.pt")
        proj_path = os.path.join(model_path, "model_proj.pt")
        if not os.path.exists(molecule_path):
            raise FileNotFoundError(f"Molecule encoder file not found: {molecule_path}")
        if not os.path.exists(proj_path):
            raise FileNotFoundError(f"Molecule projection file not found: {proj_path}")
        self.molecule_encoder.load_state_dict(torch.load(molecule_path, map_location="cpu", weights_only=False))
        self.molecule_projection.load_state_dict(torch.load(proj_path, map_location="cpu", weights_only=False))

    def save_pretrained(self, output_dir):
        os.makedirs(output_dir, exist_ok=True)
        torch.save(self.molecule_encoder.state_dict(), os.path.join(output_dir, "model.pt"))
        torch.save(self.molecule_projection.state_dict(), os.path.join(output_dir, "model_proj.pt"))
        with open(os.path.join(output_dir, "model_config.json"), "w") as f:
            json.dump(self.model_config, f, indent=2)
