"""Multi-GPU sharding of the generation path (one process per GPU, RCCL over xGMI).

Each prompt's diffusion trajectory and each molecule's A* search is independent (reference
modeling_llamole.py:1173-1190 already loops per molecule; eval/workflow.py:89-91 has no sampler), so ranks
take disjoint prompt shards with full model replicas and NO data-path collective.  The only exchange is one
small all-gather of fixed-size records per phase:

  design phase   per molecule  int8 [1 + N + N*N]  = n_nodes, atom classes, bond classes   (~1.1 KB at N=32)
  retro phase    per expansion int32[k] + f32[k]   = top-k template ids and probabilities  (400 B at k=50)

Messages are KB-scale and latency-bound, so the design issues ONE ``dist.all_gather`` per phase (never one per molecule or per
expansion); which algorithm RCCL runs under that call (direct, ring, tree) is RCCL's choice and is not set here.
``backend="nccl"`` is RCCL on ROCm; the same code runs under ``gloo`` on CPU tensors, which is how it is tested without GPUs.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def force_dist() -> bool:
    """LLAMOLE_FORCE_DIST=1: take the collective code paths even with ONE rank (process group of size 1).  The single-rank shortcuts
    below otherwise return before any collective is issued, so RCCL would meet this code for the first time on an 8-GPU node; the
    GPU test suite drives bench.py / main.py eval / the helpers below through backend "nccl" with one rank this way."""
    import os
    return os.environ.get("LLAMOLE_FORCE_DIST") == "1"


def _solo(group=None) -> bool:
    """True when there is nothing to exchange: no process group, or one rank and the collectives are not forced."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and not force_dist()


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of ``range(n_items)`` (first ``n % world`` ranks get one extra)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return range(lo, lo + q + (1 if rank < r else 0))


def pack_graphs(mols: Sequence[Tuple[torch.Tensor, torch.Tensor]], max_nodes: int, capacity: int) -> torch.Tensor:
    """[(atom_types[n], edge_types[n,n])] -> int8 [capacity, 1 + N + N*N]; unused rows have n_nodes = -1."""
    N = max_nodes
    rec = torch.full((capacity, 1 + N + N * N), -1, dtype=torch.int8)
    for b, (a, e) in enumerate(mols):
        n = int(a.numel())
        rec[b, 0] = n
        rec[b, 1:1 + n] = a.to(torch.int8)
        full = torch.full((N, N), -1, dtype=torch.int8)
        full[:n, :n] = e.to(torch.int8)
        rec[b, 1 + N:] = full.reshape(-1)
    return rec


def unpack_graphs(rec: torch.Tensor, max_nodes: int):
    N = max_nodes
    out = []
    for row in rec.cpu():
        n = int(row[0])
        if n < 0:
            continue
        out.append([row[1:1 + n].long(), row[1 + N:].reshape(N, N)[:n, :n].long()])
    return out


def all_gather_graphs(mols, max_nodes: int, n_total: int, device=None, group=None):
    """All ranks end up with the molecules of all prompts, in global prompt order."""
    if _solo(group):
        return list(mols)
    world = dist.get_world_size(group)
    cap = (n_total + world - 1) // world
    rec = pack_graphs(mols, max_nodes, cap)
    if device is not None:
        rec = rec.to(device)
    bufs = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(bufs, rec, group=group)
    out = []
    for b in bufs:
        out.extend(unpack_graphs(b, max_nodes))
    return out


def all_gather_topk(idx: torch.Tensor, prob: torch.Tensor, group=None):
    """Retro phase: gather per-rank candidate scores ([G_local,k] int32 / f32, equal G_local on every rank)."""
    return gather_rows(idx, group), gather_rows(prob, group)


def gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of equally shaped [rows, ...] tensors, rank-major ([world * rows, ...]); one collective.  Device tensors go through
    RCCL as they are under backend "nccl"; under gloo (CPU tests, single-GPU dry runs) they are staged through the host."""
    if _solo(group):
        return t
    world = dist.get_world_size(group)
    host = t.is_cuda and dist.get_backend(group) == "gloo"
    src = t.cpu() if host else t.contiguous()
    bufs = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(bufs, src, group=group)
    out = torch.cat(bufs)
    return out.to(t.device) if host else out


def allreduce_gradients(params, bucket_bytes: int = 64 << 20, group=None) -> int:
    """Average ``p.grad`` of every parameter over the ranks (SFT data parallelism): gradients are packed
    into flat buckets of up to ``bucket_bytes`` per dtype and each bucket is ONE all-reduce (RCCL over xGMI under backend
    "nccl", gloo in the CPU tests).  Llamole's trainable set (LoRA adapter + connectors) is tens of MB: one or two direct
    all-reduces, not a per-tensor stream of small ones.  Returns the number of collectives issued (0 when not distributed)."""
    if _solo(group):
        return 0
    world = dist.get_world_size(group)
    # The bucket layout must be the SAME on every rank, so it is built from the fixed list of trainable parameters, never from
    # which of them happen to carry a gradient: a rank whose shard has no <molecule> token produces no connector gradient, a
    # rank without a retro label none for lm_to_graph_predictor -- those enter the sum as zeros.  A parameter that has a gradient
    # on NO rank must stay without one, exactly as on a single GPU (AdamW would otherwise apply weight decay and moment decay to
    # it under N GPUs and skip it under one): each bucket carries one extra element per parameter, 1 where the rank has a
    # gradient, and parameters whose count comes back 0 get ``grad = None`` again.  Mismatched bucket lengths would hang or
    # corrupt an RCCL all-reduce; this layout depends on the parameter list alone.
    by_dtype = {}
    for p in params:
        if p.requires_grad:
            by_dtype.setdefault((p.dtype, p.device), []).append(p)
    calls = 0
    for (dtype, device), plist in by_dtype.items():
        bucket, size = [], 0
        itemsize = torch.empty((), dtype=dtype).element_size()

        def flush(bucket=bucket):
            nonlocal calls
            if not bucket:
                return
            total = sum(p.numel() for p in bucket)
            flat = torch.zeros(total + len(bucket), dtype=dtype, device=device)
            off = 0
            for i, p in enumerate(bucket):
                n = p.numel()
                if p.grad is not None:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                    flat[total + i] = 1
                off += n
            if flat.is_cuda and dist.get_backend(group) == "gloo":      # single-GPU dry runs of the N > 1 path: gloo reduces host tensors
                host = flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                flat.copy_(host)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            have = flat[total:].float().cpu()          # ranks holding a gradient, per parameter (<= world: exact in bf16 up to 256)
            flat[:total].div_(world)
            off = 0
            for i, p in enumerate(bucket):
                n = p.numel()
                if have[i] == 0:
                    p.grad = None
                elif p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
            calls += 1
            bucket.clear()
        for p in plist:
            nbytes = p.numel() * itemsize
            if bucket and size + nbytes > bucket_bytes:
                flush()
                size = 0
            bucket.append(p)
            size += nbytes
        flush()
    return calls


class WorkQueue:
    """Dynamic distribution of work items over the ranks (decode lengths and A* iteration counts vary from prompt
    to prompt, so a static contiguous shard can leave ranks idle): every rank claims the next unclaimed item -- a batch of
    prompts -- from ONE shared counter until the items are exhausted.  The counter lives in a ``torch.distributed.TCPStore`` on
    rank 0 (``store.add`` is atomic); no GPU collective is involved and the ranks never wait for each other.  With one rank (or
    no process group) it degenerates to ``range(n_items)``."""

    def __init__(self, n_items: int, rank: int = 0, world: int = 1, key: str = "llamole_work", port_offset: int = 17,
                 store=None, port: Optional[int] = None, timeout_s: Optional[float] = None):
        """``port``: TCP port of the counter's store (default: env LLAMOLE_QUEUE_PORT, else MASTER_PORT + ``port_offset``);
        ``store``: an existing ``torch.distributed`` store to use instead (e.g. a ``PrefixStore`` over the process group's)."""
        self.n, self.rank, self.world, self.key = int(n_items), rank, world, key
        self._local = 0
        self.store = store
        import os
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get("LLAMOLE_QUEUE_TIMEOUT_S", 1800.0))
        if world > 1 and store is None:
            import datetime
            import os
            host = os.environ.get("MASTER_ADDR", "127.0.0.1")
            if port is None:
                port = int(os.environ.get("LLAMOLE_QUEUE_PORT", int(os.environ.get("MASTER_PORT", "29500")) + port_offset))
            self.store = dist.TCPStore(host, int(port), world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=self.timeout_s),
                                       wait_for_workers=True)

    def __iter__(self):
        return self

    def __next__(self) -> int:
        if self.store is None:
            i = self._local
            self._local += 1
        else:
            i = int(self.store.add(self.key, 1)) - 1
        if i >= self.n:
            self._finish()
            raise StopIteration
        return i

    def _finish(self):
        """The store lives in rank 0's process: rank 0 leaves only after every rank has seen the end of the queue -- or raises
        after ``timeout_s`` (a peer that died mid-evaluation never reports; waiting for it forever would hang the job)."""
        if self.store is None or getattr(self, "_finished", False):
            return
        self._finished = True
        import time
        self.store.add(self.key + "_done", 1)
        if self.rank == 0:
            # the deadline counts from the last sign of life, not from the moment rank 0 ran dry: a peer still working on a long last
            # item (design + planning + rollbacks) bumps the heartbeat key from `beat()`; only `timeout_s` of silence is a dead peer
            last_seen = (int(self.store.add(self.key + "_done", 0)), int(self.store.add(self.key + "_beat", 0)))
            deadline = time.monotonic() + self.timeout_s
            while last_seen[0] < self.world:
                now = (int(self.store.add(self.key + "_done", 0)), int(self.store.add(self.key + "_beat", 0)))
                if now != last_seen:
                    last_seen, deadline = now, time.monotonic() + self.timeout_s
                elif time.monotonic() > deadline:
                    raise RuntimeError(f"WorkQueue: only {now[0]} of {self.world} ranks reached the end of the queue and none has "
                                       f"reported progress for {self.timeout_s:.0f} s -- a peer has died or hangs")
                time.sleep(0.01)

    def beat(self):
        """Sign of life from a rank that is still working on an item it claimed (called by the eval driver between the phases of a batch)."""
        if self.store is not None:
            self.store.add(self.key + "_beat", 1)
