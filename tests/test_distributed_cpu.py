"""world_size-2 gloo test of the multi-GPU sharding path (runs on CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from llamole_amd import distributed as D


def _mols(n_total, N):
    g = torch.Generator().manual_seed(0)
    out = []
    for i in range(n_total):
        n = int(torch.randint(2, N + 1, (1,), generator=g))
        a = torch.randint(0, 16, (n,), generator=g)
        e = torch.randint(0, 5, (n, n), generator=g)
        e = torch.triu(e, 1)
        out.append([a, e + e.t()])
    return out


def _worker(rank, world, port, n_total, N, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allm = _mols(n_total, N)
        mine = [allm[i] for i in D.shard_range(n_total, rank, world)]
        got = D.all_gather_graphs(mine, N, n_total)
        ok = len(got) == n_total and all(torch.equal(a, b[0]) and torch.equal(e, b[1]) for (a, e), b in zip(got, allm))
        idx = torch.full((3, 5), rank, dtype=torch.int32)
        pr = torch.full((3, 5), float(rank))
        gi, gp = D.all_gather_topk(idx, pr)
        ok = ok and gi.shape == (3 * world, 5) and gi[3 * rank, 0] == rank and float(gp[-1, 0]) == world - 1
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            cover = [i for r in range(w) for i in D.shard_range(n, r, w)]
            assert cover == list(range(n))
            sizes = [len(D.shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip():
    m = _mols(5, 12)
    rec = D.pack_graphs(m, 12, 8)
    assert rec.shape == (8, 1 + 12 + 144) and rec.dtype == torch.int8 and int(rec[5, 0]) == -1
    back = D.unpack_graphs(rec, 12)
    assert len(back) == 5 and all(torch.equal(a, b[0]) and torch.equal(e, b[1]) for (a, e), b in zip(back, m))


def test_all_gather_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, 10, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
