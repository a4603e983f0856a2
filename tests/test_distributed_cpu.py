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


# ------------------------------------------------------------------------------------------ SFT data parallelism (8 f4)
def _sft_setup():
    """Tiny real HF LLM + differentiable stand-ins for the graph engines (the HIP predictor needs a GPU); 4 samples."""
    import types
    from llamole_amd import e2e
    from llamole_amd.graph_data import GraphData
    from llamole_amd.modeling_llamole import SPECIAL_TOKENS, GraphLLMForCausalMLM
    from llamole_amd.sft import GraphSFTCollator
    llm = e2e.build_llm("tiny", "cpu", torch.float32, seed=0)
    for p in llm.parameters():
        p.requires_grad = True
    tid = {t: 2000 + i for i, t in enumerate(SPECIAL_TOKENS)}
    W = torch.randn(7, 768, generator=torch.Generator().manual_seed(3)) * 0.05

    class Pred(torch.nn.Module):
        text_input_size, available = 768, None

        def forward(self, x, ei, ea, batch, c):
            return c.float() @ W.t()
    enc = lambda x, ei, ea, b: torch.stack([x[b == g].float().mean().repeat(32) for g in range(int(b.max()) + 1)]) * 0.01  # noqa: E731
    enc.hidden_size = 32
    m = GraphLLMForCausalMLM(types.SimpleNamespace(), types.SimpleNamespace(loss_weight_lm=1, loss_weight_design=1, loss_weight_retro=1),
                             types.SimpleNamespace(learned_query_size=8), llm, types.SimpleNamespace(text_input_size=768), Pred(), enc,
                             tid, None)
    m.graph_encoder = enc
    for nm in ("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor"):
        for p in getattr(m, nm).parameters():
            p.data = torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.03
    mk = lambda n, k: GraphData(torch.arange(n) % 9 + k, torch.empty((2, 0), dtype=torch.long), torch.empty((0,), dtype=torch.long))  # noqa: E731
    graphs = {i: mk(3 + i, i) for i in range(6)}
    g = torch.Generator().manual_seed(5)
    feats = []
    for i in range(4):
        L = 30 + 2 * i
        ids = torch.randint(5, 1000, (L,), generator=g).tolist()
        ids[2] = tid["<molecule>"]
        ids[12] = tid["<retro_start>"]
        ids[13:21] = [tid["<retro_body>"]] * 8
        feats.append({"input_ids": ids, "labels": [-100] * 4 + ids[4:], "molecule_ids": [i], "retro_product_ids": [i + 1],
                      "retro_labels": [i % 7]})
    return m, GraphSFTCollator(0, graphs), feats


def _sft_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from llamole_amd.sft import sft_step
        torch.set_num_threads(2)
        m, coll, feats = _sft_setup()
        mine = [feats[i] for i in D.shard_range(len(feats), rank, world)]
        log = sft_step(m, coll(mine), optimizer=None, bucket_bytes=1 << 20)
        grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        q.put((rank, log["loss"], {k: grads[k].numpy() for k in ("lm_to_graph_predictor.0.weight", "language_model.model.layers.1.mlp.down_proj.weight",
                                                          "graph_to_lm_connector.0.bias")}))
    finally:
        dist.destroy_process_group()


def test_sft_collator_and_data_parallel_step_world2_gloo():
    """Collator semantics (reference data/collator.py:77-135) and the DP step: after the bucketed all-reduce both ranks hold
    the mean of the per-shard gradients, equal to what one process computes from the two shards."""
    from llamole_amd.sft import sft_step
    m, coll, feats = _sft_setup()
    b = coll(feats)
    assert b["input_ids"].shape == (4, 36) and b["attention_mask"].sum().item() == sum(len(f["input_ids"]) for f in feats)
    assert b["labels"][0, -1].item() == -100 and b["retro_labels"].tolist() == [[0], [1], [2], [3]]
    assert b["molecule_graphs"].num_graphs == 4 and b["retro_product_graphs"].num_graphs == 4 and b["design_graphs"].num_graphs == 4
    torch.set_num_threads(2)
    ref = {}
    for r in range(2):
        mm, cc, ff = _sft_setup()
        sft_step(mm, cc([ff[i] for i in D.shard_range(4, r, 2)]))
        for n, p in mm.named_parameters():
            if p.grad is not None:
                ref[n] = ref.get(n, 0) + p.grad / 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sft_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for k in res[0][2]:
        a, b2 = torch.from_numpy(res[0][2][k]), torch.from_numpy(res[1][2][k])
        torch.testing.assert_close(a, b2, rtol=0, atol=0)                              # ranks agree bit for bit
        torch.testing.assert_close(a, ref[k], rtol=1e-5, atol=1e-7)


def _uneven_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        a = torch.nn.Parameter(torch.ones(5))
        b = torch.nn.Parameter(torch.ones(3, 2))          # the "connector": only rank 0's shard reaches it
        c = torch.nn.Parameter(torch.ones(4))
        frozen = torch.nn.Parameter(torch.ones(2), requires_grad=False)
        unused = torch.nn.Parameter(torch.ones(3))        # trainable, but no rank's loss reaches it (a step without a <molecule> token)
        loss = (a * (rank + 1)).sum() + (c * 2).sum() + ((b * 3).sum() if rank == 0 else 0)
        loss.backward()
        assert (b.grad is None) == (rank == 1)
        n = D.allreduce_gradients([a, b, unused, c, frozen], bucket_bytes=16)     # tiny buckets: several collectives, same on both ranks
        q.put((rank, n, a.grad.tolist(), b.grad.tolist(), c.grad.tolist(), frozen.grad is None, unused.grad is None))
    finally:
        dist.destroy_process_group()


def test_allreduce_gradients_with_rank_dependent_grad_sets():
    """a rank whose shard never touches a trainable parameter (no <molecule> token -> no connector gradient) must
    still take part in that parameter's bucket, with zeros; the layout comes from requires_grad, not from grad presence."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert res[0][1] == res[1][1] >= 2
    for r in res:
        assert r[2] == [1.5] * 5 and r[3] == [[1.5, 1.5]] * 3 and r[4] == [2.0] * 4 and r[5]
        assert r[6], "a parameter without a gradient on ANY rank must keep grad = None (as on one GPU), not get zeros"


def _queue_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    wq = D.WorkQueue(11, rank, world)
    got = []
    for i in wq:
        got.append(i)
        time.sleep(0.02 if rank == 0 else 0.005)        # rank 1 is "faster": it must end up with more items
    q.put((rank, got))


def test_work_queue_hands_out_every_item_once():
    """dynamic distribution of prompt batches -- every item is claimed exactly once, the faster rank takes more."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_queue_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert sorted(res[0] + res[1]) == list(range(11)) and set(res[0]).isdisjoint(res[1])
    assert len(res[1]) > len(res[0])
    assert list(D.WorkQueue(4)) == [0, 1, 2, 3]            # one rank: plain range


def _dying_peer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    wq = D.WorkQueue(4, rank, world, timeout_s=3.0)
    if rank == 1:
        next(wq)            # claims one item, then "dies" without ever reaching the end of the queue
        q.put((rank, "left"))
        import time
        time.sleep(8)       # keeps the process (not the queue) alive so that rank 0's store connection stays valid
        return
    try:
        list(wq)
        q.put((rank, "finished"))
    except RuntimeError as e:
        q.put((rank, "raised: " + str(e)))


def test_work_queue_rank0_raises_instead_of_hanging_when_a_peer_dies():
    """rank 0 used to spin forever in _finish when a peer died mid-evaluation."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dying_peer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert res[1] == "left" and res[0].startswith("raised") and "1 of 2 ranks" in res[0], res


# ------------------------------------------------------------------------------------------------ expansion-level split
def _sharded_world(rank, world):
    """A scripted orchestrator whose every device-side answer is a function of the request alone (never of its batch mates), so that
    the sharded and the unsharded run must agree exactly: the LM 'decodes' tokens derived from the prompt embedding, its forward is a
    masked running mean, the predictor's top-k comes from a digest of (graph, condition)."""
    import types
    import zlib
    from llamole_amd.graph_data import GraphData
    from llamole_amd.modeling_llamole import GraphLLMForCausalMLM, make_connector
    from tests import host_fakes as hf

    class RowLM(hf.FakeLM):
        def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, position_ids=None, **kw):
            h = self.emb(input_ids) if inputs_embeds is None else inputs_embeds
            m = torch.ones(h.shape[:2]) if attention_mask is None else attention_mask[:, -h.shape[1]:].float()
            h = torch.cumsum(h * m[..., None], dim=1) / torch.cumsum(m, dim=1).clamp_min(1.0)[..., None]
            return types.SimpleNamespace(logits=self.head(h), hidden_states=(h, h))

        def generate(self, inputs=None, attention_mask=None, inputs_embeds=None, max_new_tokens=4, **kw):
            src = inputs_embeds if inputs_embeds is not None else self.emb(inputs)
            m = torch.ones(src.shape[:2]) if attention_mask is None else attention_mask.float()
            key = (src * m[..., None]).sum(dim=(1, 2))
            n = min(int(max_new_tokens), 6)
            new = torch.stack([(torch.arange(n) * 3 + int(abs(float(k)) * 1000) % 11) % 7 + 20 for k in key])
            return new if inputs is None else torch.cat([inputs, new], dim=1)

    class TopkPredictor(hf.FakePredictor):
        label_to_template = {i: f"T{i}" for i in range(64)}

        def topk_templates_batch(self, graphs, c, topk):
            idx, prob = [], []
            for g, row in zip(graphs, c):
                h = zlib.crc32(bytes(g.x.tolist())) ^ (int(abs(float(row.float().sum())) * 997) & 0xffff)
                idx.append([(h + 7 * k) % 64 for k in range(topk)])
                p = torch.softmax(torch.arange(topk, 0, -1).float() * 0.3, 0)
                prob.append(p.tolist())
            return torch.tensor(prob), torch.tensor(idx, dtype=torch.int32)

        def sample_templates_batch(self, graphs, c, smiles_list, topk):
            p, i = self.topk_templates_batch(graphs, c, topk)
            return self.merge_topk(p.numpy(), i.numpy(), smiles_list)

        def merge_topk(self, probs, idx, smiles_list):
            out = []
            for pr, ix, s in zip(probs, idx, smiles_list):
                d = int(s[1]) if s[0] == "P" else 0
                seen, reactants, scores, temps = {}, [], [], []
                for p, i in zip(pr, ix):
                    h = zlib.crc32(f"T{int(i)}{s}".encode())
                    r = ".".join(sorted([f"B{h % 5}" if (h >> 8) % 4 <= d else f"P{d + 1}_{h % 97}", f"B{(h >> 4) % 5}"]))
                    if r in seen:
                        scores[seen[r]] += float(p)
                    else:
                        seen[r] = len(reactants)
                        reactants.append(r); scores.append(float(p)); temps.append(f"T{int(i)}")
                tot = sum(scores)
                out.append((reactants, [v / tot for v in scores], temps))
            return out

    m = GraphLLMForCausalMLM(types.SimpleNamespace(), types.SimpleNamespace(), types.SimpleNamespace(learned_query_size=8), RowLM(),
                             hf.FakeDecoder([]), TopkPredictor({}, []), hf.FakeEncoder(), dict(hf.TOKEN_IDS), hf.Tok())
    for k, v in hf.seeded_connectors(make_connector).items():
        setattr(m, k, v)
    m.smiles_to_graph = hf.fake_smiles_to_graph(GraphData)
    m.expected_cost_value = True                      # every value forward counts (not the constant 15 of the reference's broadcast)
    m.retro_max_new_tokens = 6
    return m


def _plan(m, n_targets):
    kw = dict(expansion_topk=6, iterations=4, starting_mols={f"B{i}" for i in range(5)}, max_planning_time=1e9, rollback=False,
              design_text="Design", max_new_tokens=6)
    targets = [f"P0_{i}" for i in range(n_targets)]
    out = m.retrosynthesize_many([None] * n_targets, targets, **kw)
    return [(o["success"], o["route_length"], o["reaction_list"], None if o["cost"] is None else [round(c, 9) for c in o["cost"]],
             o["templates"]) for o in out]


def _split_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        solo = _plan(_sharded_world(rank, world), 7)                 # every rank alone, no exchange
        m = _sharded_world(rank, world)
        m.expansion_shard = (rank, world, None)
        gen, fwd = m.language_model.generate_calls, m.language_model.forward_calls
        split = _plan(m, 7)
        q.put((rank, solo == split, sum(o[0] for o in solo), len(gen), len(fwd)))
    finally:
        dist.destroy_process_group()


def test_expansion_level_split_world2_equals_solo():
    """A lock-step round's expansions and value prompts split over two ranks with a replicated host A*: one all-gather of
    (topk_idx, topk_prob) + analysis tokens per round, one of the costs per value call -- routes, costs and templates identical to
    the unsharded searches on every rank (SURVEY.md 8e second alternative; modeling_llamole.one_step_reaction_batch)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
    assert [r[1] for r in res] == [True, True], res
    assert res[0][2] >= 1 and res[0][2] == res[1][2]                 # some searches succeed, the same ones on both ranks


def test_work_queue_heartbeat_is_harmless_without_a_store():
    q = D.WorkQueue(3)
    q.beat()
    assert list(q) == [0, 1, 2]


def test_gather_rows_and_topk_are_identity_without_a_group():
    t = torch.arange(12).view(3, 4)
    assert D.gather_rows(t) is t
    i, p = D.all_gather_topk(t.int(), t.float())
    assert torch.equal(i, t.int()) and torch.equal(p, t.float())


def _diverge_worker(rank, world, port, q):
    """Rank 1 'decides' differently from rank 0 (one live request fewer): the agreement check must raise on BOTH ranks, not hang one of
    them in the gather; and the shared clock hands every rank rank 0's reading."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _sharded_world(rank, world)
        m.expansion_shard = (rank, world, None)
        t = m._shared_clock()
        tt = torch.tensor([t], dtype=torch.float64)
        both = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(both, tt)
        same_clock = bool(both[0] == both[1])
        m._assert_ranks_agree(5, None, "requests")                   # agreement: silent
        try:
            m._assert_ranks_agree(5 - rank, None, "requests")
            raised = False
        except RuntimeError as e:
            raised = "disagree" in str(e)
        q.put((rank, same_clock, raised))
    finally:
        dist.destroy_process_group()


def test_expansion_split_detects_diverged_ranks_and_shares_one_clock():
    """the expansion split issues collectives from inside a replicated host A*.  max_planning_time is judged by ONE clock (rank
    0's, broadcast per round) and a rank whose request count differs raises on every rank instead of hanging its peers."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_diverge_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True, True), (1, True, True)], res


def test_molstar_many_reads_the_clock_once_per_round():
    """All searches of a lock-step run stop in the same round when the (shared) clock runs out -- none of them a round earlier."""
    from llamole_amd.planner import molstar_many
    ticks = {"n": 0}

    def clock():
        ticks["n"] += 1
        return float(ticks["n"])            # one "second" per reading: t0 = 1, round k is judged at 1 + k

    rounds = []

    def expand(picks):
        rounds.append([i for i, _ in picks])
        return [{"reactants": [f"X{mol}a.X{mol}b"], "scores": [0.9], "templates": ["T"], "analysis": [1]} for _, mol in picks]

    out = molstar_many(["A", "B", "C"], {"Z"}, expand, value_fn=lambda s, r: 1.0, iterations=50, max_time=2.5, clock=clock)
    assert rounds == [[0, 1, 2], [0, 1, 2]]                          # rounds judged at elapsed 1 and 2 run, the one at 3 > 2.5 does not
    assert all(not ok for ok, _, _ in out) and ticks["n"] == 4
