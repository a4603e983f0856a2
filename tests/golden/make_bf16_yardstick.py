"""How far do the REFERENCE's own modules in bf16 sit from their f32 selves?  (missing #5.)

The reference's GPU path is bf16 end to end (loader.py:245-247 casts every parameter; the transition tables are built in
``model_dtype``, diffusion_model.py:78-101; timesteps are ``type_as(y)``, :280-283).  The repo's bf16 engine is held to the f32
oracle with bounds the builder chose; this script measures the yardstick those bounds should be read against: the reference
``GraphDiT`` constructed with ``model_dtype=torch.bfloat16`` and cast like its loader casts it, against the same class in f32 with
the same (bf16-representable) parameter values, on the same states and the same injected Exp(1) noise, at reverse steps
s = 49, 35, 25, 10, 3, 0 of the f32 model's own trajectory -- logit error, total variation of the guided step probabilities,
agreement of the exponential-race winners.

Run in the build container only (``python tests/golden/make_bf16_yardstick.py [fixture] [full]``); imports /root/reference with the
stubs of make_goldens.py (CPU bf16 kernels of this torch build).  Writes data only: tests/golden/bf16_yardstick.json.
Sizes: "fixture" = the golden case dit_n32_h128 (H = 128, L = 2); "full" = the benchmarked denoiser (H = 1024, L = 28, 16 heads,
N = 32, T = 50, B = 8, ragged n_nodes) with weights drawn like bench.fast_dit_weights draws them (CPU generator here: the same
distribution as the GPU tests' weights, not the same values -- the yardstick is a statistic).
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bf16_yardstick.json")

from llamole_amd import synth  # noqa: E402
from tests.golden import make_goldens as mg  # noqa: E402

PROBE_STEPS = (49, 35, 25, 10, 3, 0)


def _weights_like_bench(cfg, max_node, seed=1234):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in synth.dit_weight_shapes(cfg, max_node).items():
        if len(shp) == 1:
            gain = k.endswith(("norm.weight", "x_embedder.1.weight"))
            sd[k] = (1.0 if gain else 0.0) + (0.1 if gain else 0.05) * torch.randn(shp, generator=g)
        elif "embedding" in k:
            sd[k] = 0.5 * torch.randn(shp, generator=g)
        else:
            sd[k] = (2.0 / (shp[0] + shp[1])) ** 0.5 * torch.randn(shp, generator=g)
    return sd


def _one_step(model, du, dtype, s, T, X_t, E_t, y, text, mask, noise):
    """The reference's sample_p_zs_given_zt with taps on the denoiser outputs and on the probabilities it samples from."""
    tap = {"logits": []}
    orig_fwd, orig_sample = model._forward, du.sample_discrete_features

    def fwd(noisy, txt, unconditioned=False):
        pred = orig_fwd(noisy, txt, unconditioned=unconditioned)
        tap["logits"] += [pred.X.detach().float().clone(), pred.E.detach().float().clone()]
        return pred

    def spy(probX, probE, node_mask, step=None, add_nose=True):
        tap["pX"], tap["pE"] = probX.detach().float().clone(), probE.detach().float().clone()
        return orig_sample(probX, probE, node_mask, step, add_nose)

    model._forward, du.sample_discrete_features = fwd, spy
    try:
        B = X_t.shape[0]
        with torch.no_grad(), mg.NoiseFeed() as feed:
            feed.push(*noise)
            s_arr = (s * torch.ones((B, 1))).to(dtype)               # generate(): s_array.type_as(y)
            _, disc = model.sample_p_zs_given_zt(s_arr / T, (s_arr + 1) / T, X_t.to(dtype), E_t.to(dtype), y.to(dtype), text.to(dtype), mask)
    finally:
        model._forward, du.sample_discrete_features = orig_fwd, orig_sample
    tap["X"], tap["E"] = disc.X.long(), disc.E.long()
    return tap


def yardstick(tag, cfg, meta, sd, B, n_nodes, seed):
    from graph_decoder import diffusion_model as dm
    from graph_decoder import diffusion_utils as du
    N, T = meta["max_node"], cfg["diffusion_steps"]
    sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}       # both models hold exactly these values
    tmp = tempfile.mkdtemp()
    synth.write_dit_dir(tmp, cfg, meta, sd)
    paths = (os.path.join(tmp, "config.yaml"), os.path.join(tmp, "data.meta.json"))
    m32 = dm.GraphDiT(*paths, torch.float32)
    m32.init_model(tmp)
    m32.eval()
    m16 = dm.GraphDiT(*paths, torch.bfloat16)
    m16.init_model(tmp)
    for p in m16.parameters():                                           # loader.py:245-247
        if p.dtype == torch.float32:
            p.data = p.data.to(torch.bfloat16)
    m16.eval()
    props, text, _ = synth.make_dit_inputs(B, seed, N)
    y = torch.where(props == -200.0, torch.tensor(float("nan")), props)
    mask = torch.arange(N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)
    um = torch.zeros(B, N, N, dtype=torch.bool)
    for b in range(B):
        n = int(n_nodes[b])
        um[b, :n, :n] = torch.triu(torch.ones(n, n, dtype=torch.bool), 1)
    noise = lambda st: synth.exp_noise(seed, st, B, N)  # noqa: E731

    # ---- the f32 model's own trajectory; keep the states its probe steps start from
    states = {}
    t0 = time.time()
    with torch.no_grad(), mg.NoiseFeed() as feed:
        feed.push(*noise(T))
        zT = du.sample_discrete_feature_noise(limit_dist=m32.limit_dist, node_mask=mask)
    X, E = zT.X, zT.E
    for s in reversed(range(T)):
        if s in PROBE_STEPS:
            states[s] = (X.clone(), E.clone())
        with torch.no_grad(), mg.NoiseFeed() as feed:
            feed.push(*noise(s))
            s_arr = s * torch.ones((B, 1))
            one_hot, _ = m32.sample_p_zs_given_zt(s_arr / T, (s_arr + 1) / T, X, E, y, text, mask)
        X, E = one_hot.X, one_hot.E
    print(f"[{tag}] f32 trajectory: {time.time() - t0:.1f} s", flush=True)

    res = {}
    for s in PROBE_STEPS:
        if s >= T:
            continue
        X_t, E_t = states[s]
        t0 = time.time()
        a = _one_step(m32, du, torch.float32, s, T, X_t, E_t, y, text, mask, noise(s))
        b = _one_step(m16, du, torch.bfloat16, s, T, X_t, E_t, y, text, mask, noise(s))
        la, lb = a["logits"], b["logits"]                                 # lx_c, le_c, lx_u, le_u
        lscale = max(float(la[0].abs().max()), float(la[1].abs().max()), 1.0)
        mx, me = mask.unsqueeze(-1), um.unsqueeze(-1)
        lerr = max(float(((lb[0] - la[0]) * mx).abs().max()), float(((lb[2] - la[2]) * mx).abs().max()),
                   float(((lb[1] - la[1]) * me).abs().max()), float(((lb[3] - la[3]) * me).abs().max())) / lscale
        tvx = (0.5 * (b["pX"] - a["pX"]).abs().sum(-1))[mask]
        tve = (0.5 * (b["pE"] - a["pE"]).abs().sum(-1))[um]
        res[str(s)] = dict(alpha_bar_s=float(m32.noise_schedule.alphas_bar[s]), logit_err_rel=lerr,
                           tv_atoms_max=float(tvx.max()), tv_atoms_mean=float(tvx.mean()),
                           tv_bonds_max=float(tve.max()), tv_bonds_mean=float(tve.mean()),
                           race_agree_atoms=float((a["X"][mask] == b["X"][mask]).float().mean()),
                           race_agree_bonds=float((a["E"][um] == b["E"][um]).float().mean()),
                           n_atoms=int(mask.sum()), n_pairs=int(um.sum()))
        print(f"[{tag}] s={s}: {res[str(s)]}  ({time.time() - t0:.1f} s)", flush=True)
    return res


if __name__ == "__main__":
    torch.set_num_threads(8)
    mg.install_stubs()
    mg.check_multinomial_is_race()
    sys.path.insert(0, mg.REF)
    which = sys.argv[1:] or ["fixture", "full"]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    if "fixture" in which:
        cfg, meta, sd, B, seed = mg.dit_case("dit_n32_h128")
        _, _, n_nodes = synth.make_dit_inputs(B, seed, meta["max_node"])
        out["fixture_dit_n32_h128"] = yardstick("fixture", cfg, meta, sd, B, n_nodes, seed)
    if "full" in which:
        cfg = synth.make_dit_config(1024, 28, 16, 50, 2.0)
        meta = synth.make_data_meta(32, 0, fixed_n_nodes=32)
        sd = _weights_like_bench(cfg, 32)
        out["full_h1024_l28"] = yardstick("full", cfg, meta, sd, 8, torch.tensor([32, 32, 17, 5, 32, 1, 29, 32]), 11)
    out["_about"] = ("reference GraphDiT in bf16 (model_dtype=bfloat16, parameters cast like loader.py:245-247) vs the same class in f32 on "
                     "the same states and noise; generated by tests/golden/make_bf16_yardstick.py")
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT)
