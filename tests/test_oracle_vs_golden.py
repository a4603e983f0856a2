"""Pin the CPU oracle (oracle/*.py) to outputs of the reference itself.

The fixtures in tests/golden/*.npz were produced by tests/golden/make_goldens.py, which runs
the reference's own GraphDiT / GraphCLIP / GNNRetrosynthsizer / CostMLP modules on CPU fp32.
Tolerances: fp32 vs fp32 on the same host -> rtol 1e-4 / atol 2e-5 for activations (summation
order may differ between the reference's module calls and the oracle's functional calls);
integer outputs (sampled graphs) must match exactly.
"""
import numpy as np
import pytest
import torch

from llamole_amd import synth
from oracle import gin_oracle as go
from oracle import graphdit_oracle as do
from tests.cases import DIT_CASES, GIN_CASES, dit_case, load_golden

RT, AT = 1e-4, 2e-5


def _close(a, b, rt=RT, at=AT):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, rtol=rt, atol=at)


def _state_from_idx(Xi, Ei):
    Xi = torch.from_numpy(Xi.astype(np.int64))
    Ei = torch.from_numpy(Ei.astype(np.int64))
    X = torch.nn.functional.one_hot(Xi.clamp_min(0), 16).float() * (Xi >= 0).unsqueeze(-1)
    E = torch.nn.functional.one_hot(Ei.clamp_min(0), 5).float() * (Ei >= 0).unsqueeze(-1)
    return X, E


@pytest.fixture(scope="module", params=list(DIT_CASES))
def dit(request):
    cfg, meta, sd, B, seed = dit_case(request.param)
    g = load_golden(request.param)
    spec = do.build_spec(cfg, meta)
    props = torch.from_numpy(g["props"])
    y = torch.where(props == -200.0, torch.tensor(float("nan")), props)
    text = torch.from_numpy(g["text"])
    n_nodes = torch.from_numpy(g["n_nodes"])
    mask = torch.arange(spec.N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)
    return dict(cfg=cfg, meta=meta, sd=sd, B=B, seed=seed, g=g, spec=spec, props=props, y=y,
                text=text, n_nodes=n_nodes, mask=mask)


def test_schedule_and_transition(dit):
    g, spec = dit["g"], dit["spec"]
    _close(spec.betas, g["betas"], 1e-6, 1e-9)
    _close(spec.alphas_bar, g["alphas_bar"], 1e-6, 1e-9)
    _close(spec.u, g["u"], 1e-6, 1e-8)
    _close(spec.x_marg, g["x_marg"], 1e-6, 1e-9)
    _close(spec.e_marg, g["e_marg"], 1e-6, 1e-9)


def test_initial_state(dit):
    spec, g = dit["spec"], dit["g"]
    X, E = do.initial_state(spec, dit["mask"], *synth.exp_noise(dit["seed"], spec.T, dit["B"], spec.N))
    Xg, Eg = _state_from_idx(g["X_T"], g["E_T"])
    assert torch.equal(X, Xg) and torch.equal(E, Eg)
    # diagonal of z_T is the all-zero vector, not class 0 (diffusion_utils.py:509-515)
    assert (g["E_T"][:, np.arange(spec.N), np.arange(spec.N)] == -1).all()


def test_denoiser_pieces(dit):
    spec, g, sd = dit["spec"], dit["g"], dit["sd"]
    X, E = _state_from_idx(g["X_T"], g["E_T"])
    t = (torch.full((dit["B"], 1), float(spec.T - 1)) + 1) / spec.T
    for tag, unc in (("c", False), ("u", True)):
        lx, le, c, hs = do.denoiser(sd, spec, X, E, dit["mask"], dit["y"], dit["text"], t, unc, return_hidden=True)
        _close(c, g[f"cvec_{tag}"])
        _close(hs[0], g[f"h0_{tag}"])
        _close(hs[1], g[f"h1_{tag}"], 2e-4, 5e-5)
        _close(lx, g[f"logX_{tag}"], 3e-4, 1e-4)
        _close(le, g[f"logE_{tag}"], 3e-4, 1e-4)


def test_one_step(dit):
    spec, g, sd = dit["spec"], dit["g"], dit["sd"]
    X, E = _state_from_idx(g["X_T"], g["E_T"])
    s = spec.T - 1
    pX, pE = do.guided_probs(sd, spec, X, E, dit["mask"], dit["y"], dit["text"], s)
    _close(pX, g["step_pX"], 1e-3, 1e-6)
    _close(pE, g["step_pE"], 1e-3, 1e-6)
    Xs, Es = do.sample_features(pX, pE, dit["mask"], *synth.exp_noise(dit["seed"], s, dit["B"], spec.N))
    Xo, Eo = do.to_onehot_masked(Xs, Es, dit["mask"])
    Xi, Ei = do.collapse(Xo, Eo, dit["mask"])
    assert np.array_equal(Xi.numpy(), g["step_X"])
    assert np.array_equal(Ei.numpy(), g["step_E"])


def test_full_trajectory(dit):
    spec, g, sd = dit["spec"], dit["g"], dit["sd"]
    noise = lambda step: synth.exp_noise(dit["seed"], step, dit["B"], spec.N)  # noqa: E731
    with torch.no_grad():
        mols, _, trace = do.generate(sd, spec, dit["props"].clone(), dit["text"], dit["n_nodes"], noise,
                                     trace_every=10)
    for s, (Xi, Ei) in trace.items():
        assert np.array_equal(Xi.numpy(), g[f"trace{s}_X"]), f"trace X diverged at step {s}"
        assert np.array_equal(Ei.numpy(), g[f"trace{s}_E"]), f"trace E diverged at step {s}"
    for i, (a, e) in enumerate(mols):
        assert np.array_equal(a.numpy(), g[f"mol{i}_atoms"])
        assert np.array_equal(e.numpy(), g[f"mol{i}_bonds"])
        n = int(dit["n_nodes"][i])
        assert a.shape == (n,) and e.shape == (n, n) and np.array_equal(e.numpy(), e.numpy().T)


@pytest.mark.parametrize("name", list(GIN_CASES))
def test_gin(name):
    L, H, out_dim, G, seed = GIN_CASES[name]
    g = load_golden(name)
    x, ei, ea, batch = synth.make_mol_graphs(G, seed)
    sd_e = synth.make_gin_weights(L, H, "encoder", seed=seed)
    sd_p = synth.make_proj_weights(H, seed)
    _close(go.gin_trunk(sd_e, L, x, ei, ea, batch), g["enc_graph"], 2e-4, 2e-4)
    emb = go.graphclip_forward(sd_e, sd_p, L, x, ei, ea, batch)
    _close(emb, g["enc_out"], 2e-4, 2e-5)
    _close(emb.norm(dim=-1), np.ones(G), 1e-5, 1e-5)
    sd_r = synth.make_gin_weights(L, H, "predictor", out_dim, seed)
    c = torch.from_numpy(g["c"])
    lg = go.predictor_forward(sd_r, L, x, ei, ea, batch, c)
    _close(lg, g["logits_c"], 3e-4, 3e-4)
    _close(go.predictor_forward(sd_r, L, x, ei, ea, batch, None), g["logits_none"], 3e-4, 3e-4)
    p, i = go.template_topk(lg, 50)
    _close(p, g["topk_p"], 1e-3, 1e-7)
    assert (i.numpy() == g["topk_i"]).mean() > 0.98      # near-ties may swap neighbours
    _close(go.cost_mlp(synth.make_cost_weights(seed), synth.make_fingerprints(4, seed)), g["cost_out"], 1e-5, 1e-6)


@pytest.mark.parametrize("name", list(DIT_CASES))
def test_training_forward_matches_reference(name):
    """GraphDiT.forward of the reference (eval mode, injected timesteps incl. t = 0 and t = T, injected forward noise):
    noisy state bit-exact, masked logits and the loss to rounding (tests/golden/<case>_train.npz, make_goldens.py dit_train)."""
    from llamole_amd import synth
    from oracle import graphdit_oracle as do
    cfg, meta, sd, B, seed = dit_case(name)
    g = load_golden(name + "_train")
    spec = do.build_spec(cfg, meta)
    x, ei, ea, batch, props, text, t_int = synth.make_dit_train_batch(meta, B, seed, spec.T)
    assert np.array_equal(t_int.numpy(), g["t_int"])
    qx, qe = synth.exp_noise(seed, spec.T + 1, B, spec.N)
    loss, (X_t, E_t, lx, le) = do.train_forward(sd, spec, x, ei, ea, batch, props, text, -200.0, t_int, qx, qe)
    Xi = X_t.argmax(-1)
    Xi[X_t.sum(-1) == 0] = -1
    Ei = E_t.argmax(-1)
    Ei[E_t.sum(-1) == 0] = -1
    assert np.array_equal(Xi.numpy().astype(np.int8), g["X_t"]) and np.array_equal(Ei.numpy().astype(np.int8), g["E_t"])
    np.testing.assert_allclose(lx.numpy(), g["pX"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(le.numpy(), g["pE"], rtol=2e-4, atol=2e-5)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
