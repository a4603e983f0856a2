"""Edge-case GPU parity of the GraphDiT engine against the CPU oracle (no goldens needed: the oracle is pinned to the
reference by tests/test_oracle_vs_golden.py).  Covers what the golden cases do not: maximum graph size (N = 64, the
64-lane MFMA attention / posterior path), head_dim 16 (generic attention), guidance off, single-node and two-node graphs,
batch 1, T = 1, and the C-ABI error behaviour."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest
import torch

from llamole_amd import synth

pytestmark = pytest.mark.gpu

CASES = {
    # name: (N, H, L, heads, T, guide, n_nodes list)
    "n64_hd64": (64, 128, 2, 2, 3, 2.0, [64, 33, 7]),
    "hd16_generic": (17, 64, 2, 4, 3, 1.7, [17, 5]),
    "no_guidance": (32, 128, 1, 4, 2, 1.0, [20, 32]),
    "tiny_graphs_T1": (32, 128, 1, 4, 1, 2.0, [1, 2, 3]),
    "batch1": (20, 128, 2, 4, 2, 3.0, [13]),
    # ref-default width at batch 1: the 64-row panel kernels (K chunks 1024 for qkv / fc1 / fc2 slabs, 256 for proj slabs)
    "h1024_one_molecule": (32, 1024, 2, 16, 2, 2.0, [32]),
}


def _build(name, dtype):
    from llamole_amd.graph_decoder import GraphDiT
    from oracle import graphdit_oracle as do
    N, H, L, heads, T, guide, nn = CASES[name]
    seed = abs(hash(name)) % 1000
    cfg = synth.make_dit_config(H, L, heads, T, guide)
    meta = synth.make_data_meta(N, seed)
    sd = synth.make_dit_weights(cfg, N, seed)
    d = tempfile.mkdtemp()
    synth.write_dit_dir(d, cfg, meta, sd)
    m = GraphDiT(os.path.join(d, "config.yaml"), os.path.join(d, "data.meta.json"), dtype)
    m.init_model(d)
    m.to("cuda")
    if dtype != torch.float32:
        for p in m.parameters():
            p.data = p.data.to(dtype)
    B = len(nn)
    props, text, _ = synth.make_dit_inputs(B, seed, N)
    n_nodes = torch.tensor(nn, dtype=torch.int64)
    return m, do, do.build_spec(cfg, meta), sd, props, text, n_nodes, seed


@pytest.mark.parametrize("name", list(CASES))
def test_f32_engine_matches_oracle_step_by_step(name):
    m, do, spec, sd, props, text, n_nodes, seed = _build(name, torch.float32)
    B, N, T = len(n_nodes), spec.N, spec.T
    mask = torch.arange(N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)
    noise = lambda st: synth.exp_noise(seed, st, B, N)  # noqa: E731
    m.begin(props, text, -200.0, n_nodes)
    m.init_state(*noise(T))
    X0, E0 = do.initial_state(spec, mask, *noise(T))
    Xi, Ei = do.collapse(X0.clone(), E0.clone(), mask)
    Ei[:, torch.arange(N), torch.arange(N)] = -1        # z_T diagonal is the all-zero vector
    gx, ge = m.get_state()
    assert torch.equal(gx.cpu().long(), Xi) and torch.equal(ge.cpu().long(), Ei)
    X, E = X0, E0
    bad = tot = 0
    for s in reversed(range(T)):
        with torch.no_grad():
            lx, le = do.denoiser(sd, spec, X, E, mask, props, text, (torch.full((B, 1), float(s)) + 1) / T, False)
            pX, pE = do.guided_probs(sd, spec, X, E, mask, props, text, s)
        glx, gle = m.denoise_logits(s)
        np.testing.assert_allclose(glx[0].cpu().numpy(), lx.numpy(), rtol=5e-3, atol=2e-3)
        np.testing.assert_allclose(gle[0].cpu().numpy(), le.numpy(), rtol=5e-3, atol=2e-3)
        gpx, gpe = m.step_probs(s)
        np.testing.assert_allclose(gpx.cpu().numpy()[mask.numpy()], pX.numpy()[mask.numpy()], rtol=1e-2, atol=1e-6)
        m.step(s, *noise(s))
        Xs, Es = do.sample_features(pX, pE, mask, *noise(s))
        X, E = do.to_onehot_masked(Xs, Es, mask)
        oX, oE = do.collapse(X.clone(), E.clone(), mask)
        gx, ge = m.get_state()
        bad += int((gx.cpu().long() != oX).sum()) + int((ge.cpu().long() != oE).sum())
        tot += oX.numel() + oE.numel()
        m.set_state(oX.to(torch.int8), oE.to(torch.int8))     # teacher forcing
    assert bad == 0, f"{bad}/{tot} sampled entries differ"
    # masks / symmetry / diagonal conventions of the final state
    for b in range(B):
        n = int(n_nodes[b])
        assert (oX[b, n:] == -1).all() and (oE[b, n:, :] == -1).all() and torch.equal(oE[b], oE[b].t())


@pytest.mark.parametrize("name", ["n64_hd64", "batch1"])
def test_bf16_engine_on_edge_shapes(name):
    m, do, spec, sd, props, text, n_nodes, seed = _build(name, torch.bfloat16)
    B, N, T = len(n_nodes), spec.N, spec.T
    mask = torch.arange(N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)
    m.begin(props, text, -200.0, n_nodes)
    m.init_state(*synth.exp_noise(seed, T, B, N))
    X0, E0 = do.initial_state(spec, mask, *synth.exp_noise(seed, T, B, N))
    with torch.no_grad():
        pX, pE = do.guided_probs(sd, spec, X0, E0, mask, props, text, T - 1)
    gpx, _ = m.step_probs(T - 1)
    tv = 0.5 * np.abs(gpx.cpu().numpy()[mask.numpy()] - pX.numpy()[mask.numpy()]).sum(-1)
    assert tv.max() <= 0.06, tv.max()
    mols, _ = m.generate_graphs(props, text, -200.0, n_nodes=n_nodes, seed=3)
    for (a, e), n in zip(mols, n_nodes):
        assert a.shape == (int(n),) and torch.equal(e, e.t()) and (e.diagonal() == 0).all() and int(a.min()) >= 0


def test_c_abi_error_behaviour():
    from llamole_amd import _lib
    lib = _lib.load()
    m, do, spec, sd, props, text, n_nodes, seed = _build("batch1", torch.float32)
    m._ensure_engine()
    h = m._handle
    assert lib.ll_dit_step(h, 0, None, None, C.c_uint64(0), None) == -3 and b"ll_dit_begin" in lib.ll_last_error()      # LL_ESTATE
    m.begin(props, text, -200.0, n_nodes)
    assert lib.ll_dit_step(h, 0, None, None, C.c_uint64(0), None) == -3 and b"state" in lib.ll_last_error()
    m.init_state(seed=1)
    assert lib.ll_dit_step(h, spec.T, None, None, C.c_uint64(0), None) == -1 and b"out of range" in lib.ll_last_error()  # LL_EINVAL
    q = torch.ones(8, device="cuda")
    assert lib.ll_dit_step(h, 0, C.c_void_p(q.data_ptr()), None, C.c_uint64(0), None) == -1
    with pytest.raises(ValueError, match="n_nodes out of range"):
        m.begin(props, text, -200.0, torch.tensor([spec.N + 1]))
    with pytest.raises(ValueError, match="expected properties"):
        m.begin(props[:, :9], text, -200.0, n_nodes)
    bad = _lib.LLDitConfig(128, 2, 4, 512, 129, 10, 2.0, 0)      # max_nodes > 128
    assert lib.ll_dit_param_count(C.byref(bad)) < 0 and b"max_nodes" in lib.ll_last_error()
    assert lib.ll_softmax_topk(C.c_void_p(q.data_ptr()), 1, 8, 65, C.c_void_p(q.data_ptr()), C.c_void_p(q.data_ptr()), None) == -1
    # a correct call still works afterwards
    m.step(spec.T - 1, seed=1)
    X, E = m.get_state()
    assert X.shape == (1, spec.N)


def test_async_trajectory_equals_sync_and_overlaps_other_work():
    """generate_graphs_async (side stream, caller not blocked) returns exactly what generate_graphs returns, while other work
    queued on the caller's stream afterwards is free to run; a second launch before .result() is refused."""
    import pytest as _pt
    m, do, spec, sd, props, text, n_nodes, seed = _build("n64_hd64", torch.bfloat16)
    torch.manual_seed(11)
    ref, _ = m.generate_graphs(props, text, -200.0, n_nodes=n_nodes, seed=77)
    torch.manual_seed(11)
    h = m.generate_graphs_async(props, text, -200.0, n_nodes=n_nodes, seed=77)
    busy = torch.randn(2048, 2048, device="cuda")
    for _ in range(5):
        busy = busy @ busy * 1e-3                      # unrelated work on the caller's stream
    with _pt.raises(RuntimeError, match="pending"):
        m.generate_graphs_async(props, text, -200.0, n_nodes=n_nodes, seed=78)
    got, nn = h.result()
    assert h.run_ms is not None and h.run_ms > 0 and torch.equal(nn, n_nodes)
    for (a, e), (a0, e0) in zip(got, ref):
        assert torch.equal(a, a0) and torch.equal(e, e0)
    torch.cuda.synchronize()
    h2 = m.generate_graphs_async(props, text, -200.0, n_nodes=n_nodes, seed=77)     # engine is free again
    assert len(h2.result()[0]) == len(n_nodes)
