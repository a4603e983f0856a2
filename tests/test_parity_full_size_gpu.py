"""Oracle parity of the engines that are actually benchmarked, at the benchmarked sizes.

  * GraphDiT, bf16 engine, ref-default denoiser (H=1024, L=28, 16 heads, N=32; reference transformer.py:27-36), B=1 and
    B=8 (BASELINE configs[1] / configs[0]), in BOTH engine modes -- the synchronous one (panel GEMMs) and the overlap mode the
    pipelined e2e bench runs (LDS-DMA ring): z_T bit-exact, per-block hidden-state drift (taps after the embedder and after
    blocks 7, 14, 28), logit error, total variation of the guided step probabilities, agreement of the exponential-race
    winners under injected noise -- all against oracle.guided_probs / oracle.denoiser on the same (bf16-rounded) weights.
  * attn_mfma_kernel against attn_generic_kernel on identical bf16 q|k|v, all four <NP,HD> instances, ragged n_nodes incl. 1.
  * GIN predictor at BASELINE configs[2] size (H=512, L=5, 180 576 templates, 16 graphs, bf16 engine, f32-output rows16
    template head) against gin_oracle: logits, top-50 probabilities and index sets.

  * (round 3) the same engine against the oracle's own 50-step trajectory at the reverse steps where the denoiser decides the
    posterior (s = 35, 25, 10, 3, 0; alpha_bar 0.2 .. 1.0), B = 1 / 8 / 16, teacher-forced per step and free-running.

Tolerances are the measured values (profiles/r4_parity_full_size.json) plus margin; they are asserted here and
quoted in DESIGN.md section 3.  The measured numbers are also written to gpurun_out/r4_parity_full_size.json.
"""
import json
import os
import tempfile
import types

import numpy as np
import pytest
import torch

from llamole_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}


def _report(key, val):
    REPORT[key] = val
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "r6_parity_full_size.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


# ------------------------------------------------------------------------------------------ GraphDiT at the benchmarked size
@pytest.fixture(scope="module")
def full_dit():
    import bench
    from oracle import graphdit_oracle as do
    args = types.SimpleNamespace(hidden=1024, depth=28, heads=16, T=50, guide=2.0, nodes=32, dtype="bf16")
    m, cfg, meta, sd = bench.build_model(args, torch.device("cuda"))
    # the engine sees the bf16-cast parameters (reference loader.py:245-247); the oracle gets the same values in f32
    sd_cpu = {k: v.detach().to(torch.bfloat16).float().cpu() for k, v in sd.items()}
    spec = do.build_spec(cfg, meta)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    return m, spec, sd_cpu, do


def _upper(n_nodes, N):
    B = len(n_nodes)
    um = torch.zeros(B, N, N, dtype=torch.bool)
    for b in range(B):
        n = int(n_nodes[b])
        um[b, :n, :n] = torch.triu(torch.ones(n, n, dtype=torch.bool), 1)
    return um


@pytest.mark.parametrize("mode", ["panel", "overlap", "fused", "overlap_default"])
@pytest.mark.parametrize("B", [1, 2, 8])
def test_graphdit_ref_default_bf16_vs_oracle(full_dit, B, mode):
    """panel / overlap: seven launches per block with the synchronous or the overlap-mode GEMMs; fused: q|k|v projection +
    attention as one launch per (sequence, head) (qkv_attn_kernel; the engine's default for batch 2..16); overlap_default: the
    overlap mode as the pipelined bench runs it (ring GEMMs + the fused launch at any batch)."""
    m, spec, sd, do = full_dit
    overlap = int(mode.startswith("overlap"))
    N, T, seed = spec.N, spec.T, 11
    props, text, _ = synth.make_dit_inputs(B, seed=seed, max_node=N)
    n_nodes = torch.tensor([32] if B == 1 else [32, 11] if B == 2 else [32, 32, 17, 5, 32, 1, 29, 32])     # B = 2: the two-panel GEMM (128 rows)
    y = torch.where(props == -200.0, torch.tensor(float("nan")), props)
    mask = torch.arange(N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)
    s = T - 1
    qT, qs = synth.exp_noise(seed, T, B, N), synth.exp_noise(seed, s, B, N)
    with torch.no_grad():
        X0, E0 = do.initial_state(spec, mask, *qT)
        t = torch.full((B, 1), float(s + 1) / T)
        ref = {}
        for name, unc in (("c", False), ("u", True)):
            lx, le, _, hs = do.denoiser(sd, spec, X0, E0, mask, y, text, t, uncond=unc, return_hidden=True)
            ref[name] = (lx, le, hs)
        pX, pE = do.guided_probs(sd, spec, X0, E0, mask, y, text, s)
        Xs, Es = do.sample_features(pX, pE, mask, *qs)
        Xi, Ei = do.collapse(*do.to_onehot_masked(Xs, Es, mask), mask)
        X0i, E0i = do.collapse(X0, E0, mask)

    m.begin(props, text, -200.0, n_nodes)
    m.set_option("overlap", overlap)
    m.set_option("fused_qkv_attn", -1 if mode == "overlap_default" else int(mode.startswith("fused")))
    try:
        m.init_state(*qT)
        X, E = m.get_state()
        offd = ~torch.eye(N, dtype=torch.bool).unsqueeze(0).expand(B, -1, -1)   # z_T's diagonal is the all-zero vector (-1 in the engine)
        assert torch.equal(X.cpu().long(), X0i) and torch.equal(E.cpu().long()[offd], E0i[offd]), "z_T differs from the oracle"
        rows = mask.unsqueeze(-1)
        drift = {}
        for tap in (0, 7, 14, 28):
            lx, le, hid = m.denoise_logits(s, tap_layer=tap)
            hid = hid.cpu()
            worst = 0.0
            for p, name in enumerate(("c", "u")):
                h_ref = ref[name][2][tap]
                scale = float((h_ref * rows).abs().max())
                worst = max(worst, float(((hid[p] - h_ref) * rows).abs().max()) / scale)
            drift[tap] = worst
        lx, le = lx.cpu(), le.cpu()
        lscale = max(float(ref["c"][1].abs().max()), float(ref["c"][0].abs().max()), 1.0)
        lerr = max(float((lx[p] - ref[n][0]).abs().max()) for p, n in enumerate(("c", "u")))
        lerr = max(lerr, max(float((le[p] - ref[n][1]).abs().max()) for p, n in enumerate(("c", "u")))) / lscale
        px, pe = m.step_probs(s)
        tvx = float((0.5 * (px.cpu() - pX).abs().sum(-1))[mask].max())
        um = _upper(n_nodes, N)
        tve = float((0.5 * (pe.cpu() - pE).abs().sum(-1))[um].max()) if um.any() else 0.0
        m.step(s, *qs)
        X, E = m.get_state()
        X, E = X.cpu().long(), E.cpu().long()
        n_x, n_e = int(mask.sum()), int(um.sum())
        agree_x = float((X[mask] == Xi[mask]).float().mean())
        agree_e = float((E[um] == Ei[um]).float().mean()) if n_e else 1.0
        assert torch.equal(E, E.transpose(1, 2))
        assert torch.equal(X[~mask], Xi[~mask])          # padding stays padding
    finally:
        m.set_option("overlap", 0)
        m.set_option("fused_qkv_attn", -1)
    rec = dict(hidden_drift_rel={str(k): v for k, v in drift.items()}, logit_err_rel=lerr, tv_atoms=tvx, tv_bonds=tve,
               race_agree_atoms=agree_x, race_agree_bonds=agree_e, n_atoms=n_x, n_pairs=n_e)
    print(f"B={B} {mode}: {rec}")
    _report(f"graphdit_B{B}_{mode}", rec)
    # bf16 operands / f32 accumulation over 28 post-norm blocks; measured values in profiles/r2_parity_full_size.json
    assert drift[0] <= 1e-2 and max(drift.values()) <= 5e-2, drift
    assert lerr <= 5e-2, lerr
    assert tvx <= 3e-2 and tve <= 3e-2, (tvx, tve)
    assert agree_x >= 0.95 and agree_e >= 0.98, (agree_x, agree_e)


# ------------------------------------------------------------------------------------------ reverse steps where the denoiser matters
# s = T-1 says little about the bf16 denoiser: with the cosine schedule alpha_bar(49) = 9e-4, so the posterior
# (z_t Q_t^T) * (p0_hat Qbar_s) (reference diffusion_utils.py:476-492) is 99.9 % prior.  alpha_bar(35) = 0.21, (25) = 0.48,
# (10) = 0.90, (3) = 0.98, (0) = 1: from there on p0_hat decides, and classifier-free guidance squares its ratio on top
# (diffusion_model.py:366-382).  The oracle runs ONE free trajectory of 8 graphs; B = 1 is graph 0 of it and B = 16 the eight
# graphs plus a permutation of them (every graph's trajectory is independent of its batch mates), so that the batch-16 engine
# -- two sequences per fused q|k|v + attention workgroup, the engine's default for batch 11..24 -- is compared with the oracle
# at a batch where that kernel is what runs.
PROBE_STEPS = (49, 35, 25, 10, 3, 0)
TRAJ_SEED = 11
TRAJ_N_NODES = [32, 32, 17, 5, 32, 1, 29, 32]
BATCH_ROWS = {1: [0], 8: list(range(8)), 16: list(range(8)) + [3, 6, 0, 5, 2, 7, 1, 4]}


@pytest.fixture(scope="module")
def oracle_traj(full_dit):
    """The f32 oracle's own free-running trajectory at the benchmarked size (B = 8, ragged n_nodes, injected Exp(1) noise):
    every step's sampled state, and at PROBE_STEPS the denoiser logits and guided probabilities behind it."""
    import time
    m, spec, sd, do = full_dit
    N, T, B0 = spec.N, spec.T, 8
    assert T == 50 and max(PROBE_STEPS) == T - 1
    props, text, _ = synth.make_dit_inputs(B0, seed=TRAJ_SEED, max_node=N)
    n_nodes = torch.tensor(TRAJ_N_NODES)
    probes = {}

    def hook(s, X, E, pX, pE, logits):
        if s in PROBE_STEPS:
            probes[s] = dict(pX=pX.clone(), pE=pE.clone(), logits=[None if l is None else l.clone() for l in logits])

    noise = lambda st: synth.exp_noise(TRAJ_SEED, st, B0, N)  # noqa: E731
    t0 = time.time()
    with torch.no_grad():
        _, _, trace = do.generate(sd, spec, props.clone(), text, n_nodes, noise, trace_every=1, step_hook=hook)
    print(f"oracle trajectory: {time.time() - t0:.1f} s for {T} steps of {B0} graphs")
    return dict(props=props, text=text, n_nodes=n_nodes, trace=trace, probes=probes, B0=B0)


def _rows_noise(step, rows, B0, N):
    qx, qe = synth.exp_noise(TRAJ_SEED, step, B0, N)
    qx = qx.view(B0, N, -1)[rows].reshape(len(rows) * N, -1).contiguous()
    qe = qe.view(B0, N * N, -1)[rows].reshape(len(rows) * N * N, -1).contiguous()
    return qx, qe


@pytest.mark.parametrize("mode", ["default", "overlap_default"])
@pytest.mark.parametrize("B", [1, 8, 16])
def test_graphdit_bf16_vs_oracle_at_informative_steps(full_dit, oracle_traj, B, mode):
    """The two engine configurations bench.py runs (`default`: what --workload graphdit and the back-to-back e2e run;
    `overlap_default`: what the pipelined e2e replays), every option at its default, teacher-forced from the oracle's own
    states at PROBE_STEPS: logits, guided probabilities (total variation) and race winners under the oracle's noise; then one
    free-running bf16 trajectory against the oracle's free-running one (fraction of equal entries every 10th step)."""
    m, spec, sd, do = full_dit
    tr = oracle_traj
    N, T, B0 = spec.N, spec.T, tr["B0"]
    rows = BATCH_ROWS[B]
    props, text, n_nodes = tr["props"][rows], tr["text"][rows], tr["n_nodes"][rows]
    mask = torch.arange(N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)
    um = _upper(n_nodes, N)
    n_x, n_e = int(mask.sum()), int(um.sum())
    m.begin(props, text, -200.0, n_nodes)
    m.set_option("overlap", int(mode == "overlap_default"))
    per_step = {}
    try:
        for s in PROBE_STEPS:
            pr = tr["probes"][s]
            if s == T - 1:
                m.init_state(*_rows_noise(T, rows, B0, N))             # z_T (bit-exact with the oracle: checked above)
            else:
                oX, oE = tr["trace"][s + 1]
                m.set_state(oX[rows].to(torch.int8), oE[rows].to(torch.int8))
            lx, le = m.denoise_logits(s)
            lx, le = lx.cpu(), le.cpu()
            ref_l = [l[rows] for l in pr["logits"]]                    # lx_c, le_c, lx_u, le_u
            lscale = max(float(ref_l[0].abs().max()), float(ref_l[1].abs().max()), 1.0)
            lerr = max(float(((lx[0] - ref_l[0]) * mask.unsqueeze(-1)).abs().max()), float(((lx[1] - ref_l[2]) * mask.unsqueeze(-1)).abs().max()),
                       float(((le[0] - ref_l[1]) * um.unsqueeze(-1)).abs().max()), float(((le[1] - ref_l[3]) * um.unsqueeze(-1)).abs().max())) / lscale
            px, pe = m.step_probs(s)
            tvx_all = (0.5 * (px.cpu() - pr["pX"][rows]).abs().sum(-1))[mask]
            tve_all = (0.5 * (pe.cpu() - pr["pE"][rows]).abs().sum(-1))[um] if n_e else torch.zeros(1)
            m.step(s, *_rows_noise(s, rows, B0, N))
            X, E = m.get_state()
            X, E = X.cpu().long(), E.cpu().long()
            oX, oE = tr["trace"][s]
            oX, oE = oX[rows], oE[rows]
            assert torch.equal(E, E.transpose(1, 2)) and torch.equal(X[~mask], oX[~mask])
            per_step[s] = dict(alpha_bar_s=float(spec.alphas_bar[s]), logit_err_rel=lerr,
                               tv_atoms_max=float(tvx_all.max()), tv_atoms_mean=float(tvx_all.mean()),
                               tv_bonds_max=float(tve_all.max()), tv_bonds_mean=float(tve_all.mean()),
                               race_agree_atoms=float((X[mask] == oX[mask]).float().mean()),
                               race_agree_bonds=float((E[um] == oE[um]).float().mean()) if n_e else 1.0)
        # free-running: same z_T, same noise, the engine's own states from there on
        m.init_state(*_rows_noise(T, rows, B0, N))
        free = {}
        for s in reversed(range(T)):
            m.step(s, *_rows_noise(s, rows, B0, N))
            if s % 10 == 0 or s == T - 1:
                X, E = m.get_state()
                oX, oE = tr["trace"][s]
                eq = int((X.cpu().long()[mask] == oX[rows][mask]).sum()) + int((E.cpu().long()[um] == oE[rows][um]).sum())
                free[s] = eq / (n_x + n_e)
    finally:
        m.set_option("overlap", 0)
    rec = dict(per_step={str(k): v for k, v in per_step.items()}, free_running_equal_frac={str(k): v for k, v in free.items()},
               n_atoms=n_x, n_pairs=n_e, mlp_kernels=m.mlp_choice())
    print(f"B={B} {mode}: " + json.dumps(rec))
    _report(f"graphdit_steps_B{B}_{mode}", rec)
    # asserted bounds = measured values (profiles/r4_parity_full_size.json) plus margin.  Measured on MI355X, worst over B and mode:
    # logits <= 0.8 % of scale at every step; TV max 3e-4 (s = 49, 35), 1.2e-3 (25), 7e-3 (10), 2.5e-2 (3), 1.7e-2 (0); TV mean
    # <= 2.6e-3; race winners under the oracle's noise: 100 % down to s = 10 (bonds 99.9 %), atoms 99.4 % at s = 3 and 97.2 % at
    # s = 0 (5 of 180: CFG squares the ratio of two bf16 denoiser outputs where the posterior is all p0_hat); free-running:
    # identical to the oracle down to s = 30, 99.98 % at 20, 99 % at 10, 82-86 % at 0 (one flipped near-tie is chaotic afterwards)
    for s, r in per_step.items():
        assert r["logit_err_rel"] <= 2e-2, (s, r)
        assert r["tv_atoms_max"] <= (2e-3 if s >= 35 else 6e-2) and r["tv_bonds_max"] <= (2e-3 if s >= 35 else 6e-2), (s, r)
        assert r["tv_atoms_mean"] <= 8e-3 and r["tv_bonds_mean"] <= 8e-3, (s, r)
        assert r["race_agree_atoms"] >= (0.999 if s >= 25 else 0.93) and r["race_agree_bonds"] >= (0.999 if s >= 25 else 0.99), (s, r)
    assert free[T - 1] == 1.0 and free[30] >= 0.995 and free[20] >= 0.98 and free[10] >= 0.95 and free[0] >= 0.6, free
    # and the yardstick those builder-chosen bounds are to be read against (round 5): the REFERENCE's own GraphDiT in bf16 (model_dtype
    # bfloat16, cast like loader.py:245-247) against itself in f32 at this size, same states and noise, sits at logits 1.7-2.2 % of scale,
    # TV max 5e-3 (s = 49) .. 0.41 (s = 0), winners 92.2 % atoms / 94.7 % bonds at s = 0 (tests/golden/bf16_yardstick.json) -- its
    # posterior runs in bf16 too; the engine (f32 posterior, bf16 only under the MFMA operands) must be within 1.5 x of that at every step
    from tests.cases import assert_within_bf16_yardstick, load_bf16_yardstick
    assert_within_bf16_yardstick(per_step, load_bf16_yardstick("full_h1024_l28"), n_x, n_e)


def test_graphdit_f32_engine_vs_oracle_at_informative_steps(full_dit, oracle_traj):
    """The f32 engine (exact f32 FMA chains; the parity engine behind the bit-exact fixture tests) at the BENCHMARKED size, on the same
    bf16-rounded weights as the oracle, teacher-forced from the oracle's states at PROBE_STEPS: integer work must agree bit for bit --
    every sampled atom and bond under the oracle's noise -- and the probabilities to f32 rounding."""
    import bench
    m_bf16, spec, sd, do = full_dit
    tr = oracle_traj
    args = types.SimpleNamespace(hidden=1024, depth=28, heads=16, T=50, guide=2.0, nodes=32, dtype="f32")
    m, _, _, _ = bench.build_model(args, torch.device("cuda"))
    m.denoiser.load_state_dict({k: v.cuda() for k, v in sd.items()})          # the oracle's (bf16-rounded) values as f32 masters
    N, T, B0 = spec.N, spec.T, tr["B0"]
    rows = BATCH_ROWS[8]
    props, text, n_nodes = tr["props"][rows], tr["text"][rows], tr["n_nodes"][rows]
    mask = torch.arange(N).unsqueeze(0).expand(8, -1) < n_nodes.unsqueeze(1)
    um = _upper(n_nodes, N)
    m.begin(props, text, -200.0, n_nodes)
    bad = total = 0
    worst_tv = worst_l = 0.0
    for s in PROBE_STEPS:
        pr = tr["probes"][s]
        if s == T - 1:
            m.init_state(*_rows_noise(T, rows, B0, N))
        else:
            oX, oE = tr["trace"][s + 1]
            m.set_state(oX[rows].to(torch.int8), oE[rows].to(torch.int8))
        lx, le = m.denoise_logits(s)
        ref_l = pr["logits"]
        lscale = max(float(ref_l[0].abs().max()), float(ref_l[1].abs().max()), 1.0)
        worst_l = max(worst_l, float(((lx[0].cpu() - ref_l[0]) * mask.unsqueeze(-1)).abs().max()) / lscale,
                      float(((le[1].cpu() - ref_l[3]) * um.unsqueeze(-1)).abs().max()) / lscale)
        px, pe = m.step_probs(s)
        worst_tv = max(worst_tv, float((0.5 * (px.cpu() - pr["pX"]).abs().sum(-1))[mask].max()),
                       float((0.5 * (pe.cpu() - pr["pE"]).abs().sum(-1))[um].max()))
        m.step(s, *_rows_noise(s, rows, B0, N))
        X, E = m.get_state()
        oX, oE = tr["trace"][s]
        bad += int((X.cpu().long()[mask] != oX[mask]).sum()) + int((E.cpu().long()[um] != oE[um]).sum())
        total += int(mask.sum()) + int(um.sum())
    rec = dict(mismatched_samples=bad, samples=total, tv_max=worst_tv, logit_err_rel=worst_l)
    print(f"f32 engine at the benchmarked size: {rec}")
    _report("graphdit_steps_B8_f32_engine", rec)
    assert bad == 0, rec                       # bit-exact integer work
    assert worst_tv <= 1e-4 and worst_l <= 1e-4, rec
    del m
    torch.cuda.empty_cache()


def test_graphdit_bf16_n50_T500_vs_oracle():
    """The reference's class defaults beyond the BASELINE shape: max_n_nodes = 50 (transformer.py:27; the 64-row attention tile, two-chunk
    q|k|v panels, the 65..224-row panel GEMMs) and T = 500 reverse steps (hoisted tables of 501 x (B+1) rows: 1.1 GB at B = 2), bf16 engine
    at H = 1024, L = 28: z_T bit-exact, then the same state handed to engine and oracle at s = 499, 498 (the oracle's own steps from z_T), 250 and
    5 (states the engine reached on the device in between): logits, guided probabilities, race winners under shared noise."""
    import bench
    from oracle import graphdit_oracle as do
    args = types.SimpleNamespace(hidden=1024, depth=28, heads=16, T=500, guide=2.0, nodes=50, dtype="bf16")
    m, cfg, meta, sd = bench.build_model(args, torch.device("cuda"))
    sd_cpu = {k: v.detach().to(torch.bfloat16).float().cpu() for k, v in sd.items()}
    spec = do.build_spec(cfg, meta)
    N, T, B, seed = spec.N, spec.T, 2, 21
    assert N == 50 and T == 500
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    props, text, _ = synth.make_dit_inputs(B, seed=seed, max_node=N)
    n_nodes = torch.tensor([50, 37])
    y = torch.where(props == -200.0, torch.tensor(float("nan")), props)
    mask = torch.arange(N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)
    um = _upper(n_nodes, N)
    m.begin(props, text, -200.0, n_nodes)
    m.init_state(*synth.exp_noise(seed, T, B, N))
    with torch.no_grad():
        X, E = do.initial_state(spec, mask, *synth.exp_noise(seed, T, B, N))
    Xi, Ei = do.collapse(X, E, mask)
    gX, gE = m.get_state()
    offd = ~torch.eye(N, dtype=torch.bool).unsqueeze(0).expand(B, -1, -1)
    assert torch.equal(gX.cpu().long(), Xi) and torch.equal(gE.cpu().long()[offd], Ei[offd])
    rec = {}
    first = True
    prev = None
    for s in (499, 498, 250, 5):
        if prev is not None and prev - s > 1:      # a later state: let the engine run on-device from prev - 1 down to s + 1, then hand ITS state to both
            for t in range(prev - 1, s, -1):
                m.step(t, None, None, seed=1000 + t)
            gX, gE = m.get_state()
            gX, gE = gX.cpu().long(), gE.cpu().long()
            X, E = do.to_onehot_masked(gX.clamp_min(0), gE.clamp_min(0), mask)
        if not first:
            cX, cE = do.collapse(X.clone(), E.clone(), mask)
            m.set_state(cX.to(torch.int8), cE.to(torch.int8))
        first = False
        prev = s
        with torch.no_grad():
            pX, pE, logits = do.guided_probs(sd_cpu, spec, X, E, mask, y, text, s, return_logits=True)
        lx, le = m.denoise_logits(s)
        lscale = max(float(logits[0].abs().max()), float(logits[1].abs().max()), 1.0)
        lerr = max(float(((lx[0].cpu() - logits[0]) * mask.unsqueeze(-1)).abs().max()), float(((le[1].cpu() - logits[3]) * um.unsqueeze(-1)).abs().max())) / lscale
        px, pe = m.step_probs(s)
        tvx = float((0.5 * (px.cpu() - pX).abs().sum(-1))[mask].max())
        tve = float((0.5 * (pe.cpu() - pE).abs().sum(-1))[um].max())
        qx, qe = synth.exp_noise(seed, s, B, N)
        with torch.no_grad():
            Xs, Es = do.sample_features(pX, pE, mask, qx, qe)
            X, E = do.to_onehot_masked(Xs, Es, mask)
        oX, oE = do.collapse(X.clone(), E.clone(), mask)
        m.step(s, qx, qe)
        gX, gE = m.get_state()
        agree_x = float((gX.cpu().long()[mask] == oX[mask]).float().mean())
        agree_e = float((gE.cpu().long()[um] == oE[um]).float().mean())
        rec[str(s)] = dict(alpha_bar_s=float(spec.alphas_bar[s]), logit_err_rel=lerr, tv_atoms_max=tvx, tv_bonds_max=tve,
                           race_agree_atoms=agree_x, race_agree_bonds=agree_e)
    print(f"N=50 T=500 B=2: {json.dumps(rec)}")
    _report("graphdit_N50_T500_B2", rec)
    for s, r in rec.items():
        assert r["logit_err_rel"] <= 2e-2 and r["tv_atoms_max"] <= 6e-2 and r["tv_bonds_max"] <= 6e-2, (s, r)
        assert r["race_agree_atoms"] >= 0.93 and r["race_agree_bonds"] >= 0.99, (s, r)
    del m
    torch.cuda.empty_cache()


def test_graphdit_training_forward_full_size_vs_oracle(full_dit):
    """GraphDiT.forward (the denoiser's training loss -- densify, forward-diffuse every graph to its own timestep incl. t = 0
    and t = T, one conditional pass with per-graph table rows through ll_dit_denoise_rows, masked cross-entropies) on the BENCHMARKED bf16
    engine against oracle.train_forward on the same bf16-rounded weights: noisy state bit-exact, logits and loss within bf16 tolerance."""
    m, spec, sd, do = full_dit
    B, seed = 6, 13
    meta = synth.make_data_meta(spec.N, 0, fixed_n_nodes=spec.N)
    x, ei, ea, batch, props, text, t_int = synth.make_dit_train_batch(meta, B, seed, spec.T)
    qx, qe = synth.exp_noise(seed, spec.T + 1, B, spec.N)
    loss = m(x, ei, ea, batch, props, text, -200.0, t_int=t_int, noise=(qx, qe))
    with torch.no_grad():
        ref_loss, (X_t, E_t, lx, le) = do.train_forward(sd, spec, x, ei, ea, batch, props, text, -200.0, t_int, qx, qe)
    lt = m._last_train
    counts = torch.bincount(batch, minlength=B)
    mask = torch.arange(spec.N).unsqueeze(0) < counts.unsqueeze(1)
    um = _upper(counts, spec.N)
    oX, oE = (X_t.argmax(-1), E_t.argmax(-1)) if X_t.dim() == 3 else (X_t.long(), E_t.long())
    assert torch.equal(lt["X_t"].cpu().long()[mask], oX[mask]) and torch.equal(lt["E_t"].cpu().long()[um], oE[um]), "noisy state differs from the oracle"
    lscale = max(float(lx.abs().max()), float(le.abs().max()), 1.0)
    lerr = max(float(((lt["logX"].cpu() - lx) * mask.unsqueeze(-1)).abs().max()), float(((lt["logE"].cpu() - le) * um.unsqueeze(-1)).abs().max())) / lscale
    rel = abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))
    rec = dict(loss=float(loss), loss_ref=float(ref_loss), loss_rel=rel, logit_err_rel=lerr, t_int=t_int.view(-1).tolist())
    print(f"training forward at the benchmarked size: {rec}")
    _report("graphdit_train_forward_full_size", rec)
    assert lerr <= 2e-2 and rel <= 1e-2, rec


# ------------------------------------------------------------------------------------------ MFMA attention vs f32-LDS attention
@pytest.mark.parametrize("N,H,heads", [(32, 128, 4), (32, 256, 4), (50, 128, 4), (50, 256, 4)],
                         ids=["NP32_HD32", "NP32_HD64", "NP64_HD32", "NP64_HD64"])
def test_attn_mfma_vs_generic_on_identical_qkv(N, H, heads):
    """One block of the bf16 engine run twice on the same state: attn_mfma_kernel<NP,HD> and attn_generic_kernel<bf16> read the
    same bf16 q|k|v buffer (same x, same qkv GEMM); the hidden state after block 1 may differ only by the bf16 rounding of the
    softmax probabilities the MFMA kernel feeds to P.V (the generic kernel keeps P in f32)."""
    from llamole_amd.graph_decoder import GraphDiT
    seed, B = 5, 4
    cfg = synth.make_dit_config(H, 2, heads, 10, 2.0)
    meta = synth.make_data_meta(N, seed)
    sd = synth.make_dit_weights(cfg, N, seed)
    d = tempfile.mkdtemp()
    synth.write_dit_dir(d, cfg, meta, sd)
    m = GraphDiT(os.path.join(d, "config.yaml"), os.path.join(d, "data.meta.json"), torch.bfloat16)
    m.init_model(d)
    m.to("cuda")
    for p in m.parameters():
        p.data = p.data.to(torch.bfloat16)
    props, text, _ = synth.make_dit_inputs(B, seed, N)
    n_nodes = torch.tensor([N, 1, max(2, N // 2 + 1), N - 1])
    m.begin(props, text, -200.0, n_nodes)
    m.init_state(*synth.exp_noise(seed, m.T, B, N))
    s = m.T - 1
    out = {}
    for generic in (0, 1):
        m.set_option("generic_attn", generic)
        lx, le, h1 = m.denoise_logits(s, tap_layer=1)
        out[generic] = (h1.cpu(), lx.cpu(), le.cpu())
    m.set_option("generic_attn", 0)
    valid = (torch.arange(N).unsqueeze(0) < n_nodes.unsqueeze(1)).unsqueeze(0).unsqueeze(-1)      # [1,B,N,1]
    scale = float((out[1][0] * valid).abs().max())
    err = float(((out[0][0] - out[1][0]) * valid).abs().max()) / scale
    lerr = float((out[0][1] - out[1][1]).abs().max()) / max(1.0, float(out[1][1].abs().max()))
    print(f"attn mfma vs generic N={N} hd={H // heads}: hidden {err:.3e} logits {lerr:.3e}")
    _report(f"attn_mfma_vs_generic_N{N}_hd{H // heads}", dict(hidden_rel=err, logits_rel=lerr))
    assert err <= 1e-2 and lerr <= 1e-2, (err, lerr)


# ------------------------------------------------------------------------------------------ fused q|k|v + attention launch
@pytest.mark.parametrize("N,H,heads,B", [(32, 256, 4, 3), (50, 256, 4, 4), (50, 1024, 16, 2), (20, 512, 8, 5), (32, 1024, 16, 3),
                                         (24, 2048, 32, 3)],
                         ids=["NP32_H256", "NP64_H256", "NP64_H1024_two_chunks", "N20_H512", "NP32_H1024", "H2048_two_chunks"])
def test_fused_qkv_attention_vs_separate_launches(N, H, heads, B):
    """qkv_attn_kernel (packed q|k|v weight straight into MFMA operands, token panel in LDS, attention on the LDS image of
    q|k|v) against the q|k|v GEMM + attn_mfma_kernel pair on the same state: both round q|k|v to bf16 before the per-head
    LayerNorm, so they differ only by the accumulation order of the projection."""
    from llamole_amd.graph_decoder import GraphDiT
    seed = 7
    cfg = synth.make_dit_config(H, 2, heads, 10, 2.0)
    meta = synth.make_data_meta(N, seed)
    sd = synth.make_dit_weights(cfg, N, seed)
    d = tempfile.mkdtemp()
    synth.write_dit_dir(d, cfg, meta, sd)
    m = GraphDiT(os.path.join(d, "config.yaml"), os.path.join(d, "data.meta.json"), torch.bfloat16)
    m.init_model(d)
    m.to("cuda")
    for p in m.parameters():
        p.data = p.data.to(torch.bfloat16)
    props, text, _ = synth.make_dit_inputs(B, seed, N)
    n_nodes = torch.tensor(([N, 1, max(2, N // 2 + 1), N - 1, N])[:B])
    m.begin(props, text, -200.0, n_nodes)
    m.init_state(*synth.exp_noise(seed, m.T, B, N))
    s = m.T - 1
    out = {}
    modes = (0, 1, 2) if N <= 32 and H % 512 == 0 else (0, 1)      # 2: two sequences per workgroup (graphs of <= 32 nodes)
    for fused in modes:
        m.set_option("fused_qkv_attn", fused)
        lx, le, h = m.denoise_logits(s, tap_layer=2)
        out[fused] = (h.cpu(), lx.cpu(), le.cpu())
    for fused in modes[1:]:          # run-to-run determinism of the fused launch (its waves hand over through LDS counters, not barriers)
        m.set_option("fused_qkv_attn", fused)
        lx, le, h = m.denoise_logits(s, tap_layer=2)
        assert torch.equal(h.cpu(), out[fused][0]) and torch.equal(lx.cpu(), out[fused][1]) and torch.equal(le.cpu(), out[fused][2])
    m.set_option("fused_qkv_attn", -1)
    valid = (torch.arange(N).unsqueeze(0) < n_nodes.unsqueeze(1)).unsqueeze(0).unsqueeze(-1)      # [1,B,N,1]
    assert torch.isfinite(out[1][0]).all() and torch.isfinite(out[1][1]).all() and torch.isfinite(out[1][2]).all()
    scale = float((out[0][0] * valid).abs().max())
    err = float(((out[0][0] - out[1][0]) * valid).abs().max()) / scale
    lerr = max(float((out[0][1] - out[1][1]).abs().max()), float((out[0][2] - out[1][2]).abs().max()))
    lerr /= max(1.0, float(out[0][1].abs().max()), float(out[0][2].abs().max()))
    if 2 in out:
        assert torch.isfinite(out[2][0]).all() and torch.isfinite(out[2][1]).all() and torch.isfinite(out[2][2]).all()
        err2 = float(((out[0][0] - out[2][0]) * valid).abs().max()) / scale
        lerr2 = max(float((out[0][1] - out[2][1]).abs().max()), float((out[0][2] - out[2][2]).abs().max()))
        lerr2 /= max(1.0, float(out[0][1].abs().max()), float(out[0][2].abs().max()))
        print(f"fused qkv+attn (two sequences per workgroup) vs separate N={N} H={H}: hidden {err2:.3e} logits {lerr2:.3e}")
        _report(f"fused_pair_qkv_attn_N{N}_H{H}", dict(hidden_rel=err2, logits_rel=lerr2))
        assert err2 <= 1e-2 and lerr2 <= 1e-2, (err2, lerr2)
    print(f"fused qkv+attn vs separate N={N} H={H}: hidden {err:.3e} logits {lerr:.3e}")
    _report(f"fused_qkv_attn_N{N}_H{H}", dict(hidden_rel=err, logits_rel=lerr))
    assert err <= 1e-2 and lerr <= 1e-2, (err, lerr)


def test_panel_gemm_on_packed_weights_is_bit_identical(full_dit):
    """Batch 1 at the benchmarked size: gemm_m64_kernel reading its weight fragments from the engine's MFMA-operand-order copies
    (default) against the same kernel on the row-major weights -- same products in the same order."""
    from llamole_amd import _lib
    m, spec, _, _ = full_dit
    lib = _lib.load()
    B, N, seed = 1, spec.N, 3
    props, text, _ = synth.make_dit_inputs(B, seed=seed, max_node=N)
    m.begin(props, text, -200.0, torch.tensor([N - 3]))
    m.init_state(*synth.exp_noise(seed, spec.T, B, N))
    out = {}
    old = lib.ll_set_m64_packed(1)
    try:
        for packed in (1, 0):
            lib.ll_set_m64_packed(packed)
            lx, le, h = m.denoise_logits(spec.T - 1, tap_layer=28)
            out[packed] = (lx.clone(), le.clone(), h.clone())
    finally:
        lib.ll_set_m64_packed(old)
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------ GIN predictor at configs[2] size
def test_gin_predictor_full_size_vs_oracle():
    import sys
    from llamole_amd.workloads import device_gin_weights as fast_weights
    from llamole_amd.graph_predictor import GraphPredictor
    from oracle import gin_oracle as go
    L, H, G, D, k = 5, 512, 16, 180576, 50
    dev = torch.device("cuda")
    x, ei, ea, batch = synth.make_mol_graphs(G, 0, min_atoms=32, max_atoms=32)
    pred = GraphPredictor(L, H, 0.0, D, {}, {})
    pred.to(dev)
    sdp = fast_weights(synth.gin_weight_shapes(L, H, "predictor", D), dev, 3)
    # xavier-initialised heads give near-uniform template distributions (180 k probabilities within 10 % of each other), where
    # a top-50 SET is decided by rounding noise; a trained head is peaked -- scale the last Linear so the logits span ~ +-10
    sdp["decoder.4.weight"] = sdp["decoder.4.weight"] * 40.0
    pred.predictor.load_state_dict(sdp)
    for p in pred.parameters():
        p.data = p.data.to(torch.bfloat16)
    sd_cpu = {k_: v.detach().to(torch.bfloat16).float().cpu() for k_, v in sdp.items()}
    c = torch.randn(G, 768, generator=torch.Generator().manual_seed(4))
    xs, eis, eas, bs = x.to(dev), ei.to(dev), ea.to(dev), batch.to(dev)
    logits = pred(xs, eis, eas, bs, c.to(dev)).float().cpu()
    p_gpu, i_gpu = pred.topk_templates(xs, eis, eas, bs, c.to(dev), k)
    p_gpu, i_gpu = p_gpu.cpu(), i_gpu.cpu().long()
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        ref = go.predictor_forward(sd_cpu, L, x, ei, ea, batch, c)
        rp, ri = go.template_topk(ref, k)
    scale = float(ref.abs().max())
    lerr = float((logits - ref).abs().max()) / scale
    # (forward() hands out bf16 logits like the reference's bf16 module; topk_templates runs the f32-output template head, so the
    # two are compared with the oracle separately; exactness of the top-k selection itself: test_gin_gpu.py::test_softmax_topk_*)
    assert (torch.diff(p_gpu, dim=1) <= 0).all() and int(i_gpu.min()) >= 0 and int(i_gpu.max()) < D
    overlap = float(np.mean([len(set(ri[g].tolist()) & set(i_gpu[g].tolist())) / k for g in range(G)]))
    top1 = float((ri[:, 0] == i_gpu[:, 0]).float().mean())
    # probability mass: compare the oracle's probability of the engine's picks with the oracle's own top-k mass
    ref_p = torch.softmax(ref, dim=1)
    mass_ratio = float((torch.gather(ref_p, 1, i_gpu).sum(1) / rp.sum(1)).min())
    perr = float(((p_gpu - rp).abs() / rp)[:, :10].max())       # rank-matched probabilities of the 10 likeliest templates
    rec = dict(logit_err_rel=lerr, top50_set_overlap=overlap, top1_agree=top1, top50_mass_ratio_min=mass_ratio, top50_prob_rel_err_max=perr)
    print("GIN predictor full size:", rec)
    _report("gin_predictor_full", rec)
    assert lerr <= 3e-2, lerr
    assert overlap >= 0.9 and mass_ratio >= 0.98 and top1 >= 0.9, (overlap, mass_ratio, top1)


def test_gin_backward_full_size_vs_oracle_autograd():
    """VERDICT r2 missing #6: the reverse sweep ll_gin_backward_c at BASELINE configs[4] size -- 16 product graphs of 32 atoms, H = 512,
    L = 5, 180 576 templates (the 740 MB template head read once more as dlogits x W with its 16-way split-K), bf16 engine -- against
    torch.autograd through the f32 CPU oracle on the same bf16-rounded weights: loss and d(retro cross-entropy)/d c."""
    import sys
    import torch.nn.functional as F
    from llamole_amd.workloads import device_gin_weights as fast_weights
    from llamole_amd.graph_predictor import GraphPredictor
    from oracle import gin_oracle as go
    L, H, G, D = 5, 512, 16, 180576
    dev = torch.device("cuda")
    x, ei, ea, batch = synth.make_mol_graphs(G, 1, min_atoms=32, max_atoms=32)
    pred = GraphPredictor(L, H, 0.0, D, {}, {})
    pred.to(dev)
    sdp = fast_weights(synth.gin_weight_shapes(L, H, "predictor", D), dev, 5)
    sdp["decoder.4.weight"] = sdp["decoder.4.weight"] * 40.0          # a peaked head, as a trained one (see the forward test above)
    pred.predictor.load_state_dict(sdp)
    for p in pred.parameters():
        p.data = p.data.to(torch.bfloat16)
    sd_cpu = {k_: v.detach().to(torch.bfloat16).float().cpu() for k_, v in sdp.items()}
    g = torch.Generator().manual_seed(6)
    c0 = torch.randn(G, 768, generator=g)
    labels = torch.randint(0, D, (G,), generator=g)
    c = c0.clone().to(dev).requires_grad_(True)
    logits = pred(x.to(dev), ei.to(dev), ea.to(dev), batch.to(dev), c)
    loss = F.cross_entropy(logits.float(), labels.to(dev))
    loss.backward()
    dc = c.grad.float().cpu()
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    c_ref = c0.clone().requires_grad_(True)
    loss_ref = F.cross_entropy(go.predictor_forward(sd_cpu, L, x, ei, ea, batch, c_ref), labels)
    (dc_ref,) = torch.autograd.grad(loss_ref, c_ref)
    scale = float(dc_ref.abs().max())
    err = float((dc - dc_ref).abs().max()) / scale
    cos = float(F.cosine_similarity(dc.flatten(), dc_ref.flatten(), dim=0))
    row_cos = float(F.cosine_similarity(dc, dc_ref, dim=1).min())
    lerr = abs(float(loss.detach()) - float(loss_ref.detach())) / max(1.0, abs(float(loss_ref.detach())))
    rec = dict(loss=float(loss.detach()), loss_ref=float(loss_ref.detach()), loss_rel=lerr, dc_err_rel_to_max=err, dc_cosine=cos, dc_cosine_min_row=row_cos)
    print(f"GIN reverse sweep full size: {rec}")
    _report("gin_backward_full_size", rec)
    assert scale > 0 and torch.isfinite(dc).all()
    assert lerr <= 2e-2, rec
    assert cos >= 0.995 and row_cos >= 0.98 and err <= 0.1, rec


def test_gin_encoder_full_size_vs_oracle():
    """GraphCLIP encoder at the reference's size (5 layers, hidden 512; bf16 engine) on 16 molecule graphs of 32 atoms against the f32 oracle on
    the same bf16-rounded weights: unit-norm embeddings, worst component and worst cosine."""
    import sys
    import torch.nn.functional as F
    from llamole_amd.workloads import device_gin_weights as fast_weights
    from llamole_amd.graph_encoder import GraphCLIP
    from oracle import gin_oracle as go
    L, H, G = 5, 512, 16
    dev = torch.device("cuda")
    x, ei, ea, batch = synth.make_mol_graphs(G, 2, min_atoms=32, max_atoms=32)
    sde = fast_weights(synth.gin_weight_shapes(L, H, "encoder"), dev, 1)
    sdj = fast_weights(synth.proj_weight_shapes(H), dev, 2)
    enc = GraphCLIP(L, H, 0.0, {})
    enc.to(dev)
    enc.molecule_encoder.load_state_dict(sde)
    enc.molecule_projection.load_state_dict(sdj)
    for p in enc.parameters():
        p.data = p.data.to(torch.bfloat16)
    got = enc(x.to(dev), ei.to(dev), ea.to(dev), batch.to(dev)).float().cpu()
    r = lambda d: {k: v.detach().to(torch.bfloat16).float().cpu() for k, v in d.items()}      # noqa: E731
    ref = go.graphclip_forward(r(sde), r(sdj), L, x, ei, ea, batch)
    err = float((got - ref).abs().max())
    cos = float(F.cosine_similarity(got, ref, dim=1).min())
    rec = dict(max_abs_err=err, min_cosine=cos, max_component=float(ref.abs().max()))
    print(f"GIN encoder full size: {rec}")
    _report("gin_encoder_full_size", rec)
    assert torch.allclose(got.norm(dim=1), torch.ones(G), atol=1e-2)
    assert cos >= 0.999 and err <= 2e-2, rec


def test_gin_relabelling_and_batch_order_invariance_full_size():
    """Size-independent properties of the message-passing stack at the reference's size (5 layers, hidden 512, bf16 engines; predictor with
    the 180 576-template head): relabelling the atoms of every molecule (edges renamed with them) and reordering the molecules of the batch
    leave each molecule's embedding / template logits where they were, up to the summation order of the add-aggregation and the pools
    (reference graph_encoder/model.py:124-176: scatter-add messages, segment max / sum pools; PyG semantics).  48 ragged molecules."""
    import torch.nn.functional as F
    from llamole_amd.workloads import build_gin_pair
    dev = torch.device("cuda")
    G = 48
    x, ei, ea, batch = synth.make_mol_graphs(G, 7, min_atoms=3, max_atoms=32)
    enc, pred, _ = build_gin_pair(dev, 180576, templates=False)
    c = torch.randn(G, 768, generator=torch.Generator().manual_seed(4))
    gen = torch.Generator().manual_seed(9)
    sizes = torch.bincount(batch, minlength=G).tolist()
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    # (1) new atom numbering inside every molecule; (2) new molecule order
    order = torch.randperm(G, generator=gen).tolist()
    new_of_old = torch.empty(x.shape[0], dtype=torch.long)
    xs, bs, pos = [], [], 0
    for new_g, g in enumerate(order):
        perm = torch.randperm(sizes[g], generator=gen)                  # new local index -> old local index
        xs.append(x[offs[g]:offs[g + 1]][perm])
        new_of_old[offs[g] + perm] = pos + torch.arange(sizes[g])
        bs.append(torch.full((sizes[g],), new_g, dtype=torch.long))
        pos += sizes[g]
    x2, b2 = torch.cat(xs), torch.cat(bs)
    eperm = torch.randperm(ei.shape[1], generator=gen)                   # edge list order is free as well
    ei2, ea2 = new_of_old[ei][:, eperm], ea[eperm]
    with torch.no_grad():
        e1 = enc(x.to(dev), ei.to(dev), ea.to(dev), batch.to(dev)).float().cpu()
        e2 = enc(x2.to(dev), ei2.to(dev), ea2.to(dev), b2.to(dev)).float().cpu()
        p1 = pred(x.to(dev), ei.to(dev), ea.to(dev), batch.to(dev), c.to(dev)).float().cpu()
        p2 = pred(x2.to(dev), ei2.to(dev), ea2.to(dev), b2.to(dev), c[order].to(dev)).float().cpu()
    e2b, p2b = torch.empty_like(e1), torch.empty_like(p1)
    e2b[order], p2b[order] = e2, p2
    scale = float(p1.abs().max())
    rec = dict(embedding_max_abs=float((e1 - e2b).abs().max()), embedding_min_cosine=float(F.cosine_similarity(e1, e2b, dim=1).min()),
               logits_max_abs_rel_to_scale=float((p1 - p2b).abs().max()) / scale,
               top1_equal=int((p1.argmax(dim=1) == p2b.argmax(dim=1)).sum()), graphs=G)
    print(f"GIN relabelling / batch order: {rec}")
    _report("gin_relabelling_invariance", rec)
    assert rec["embedding_min_cosine"] >= 0.9995 and rec["embedding_max_abs"] <= 1e-2, rec
    assert rec["logits_max_abs_rel_to_scale"] <= 2e-2 and rec["top1_equal"] >= G - 1, rec
