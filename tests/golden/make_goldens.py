"""Generate golden fixtures by running the REFERENCE's own modules on CPU (fp32).

Run in the build container only (``python tests/golden/make_goldens.py``): it imports
``/root/reference/src/model/graph_{decoder,encoder,predictor}`` with import-time stubs for
the third-party packages that are absent here (torch_geometric, rdkit, rdchiral), feeds
them the seeded synthetic configs / weights / inputs of ``llamole_amd.synth`` and writes
only *data* (inputs that are not re-derivable + the reference's outputs) to
``tests/golden/*.npz``.  Nothing from the reference travels: no source, no bytecode.

Stub semantics (documented because they define what the goldens pin):
  torch_geometric.nn.MessagePassing(aggr='add').propagate: x_j = x[edge_index[0]],
      message(x_j, edge_attr) summed into rows edge_index[1]; update() applied.
  global_add_pool / global_max_pool: segment sum / max over ``batch``.
  rdkit / rdchiral / torch_geometric.utils: names only (never called on the paths used).
Sampling: ``Tensor.multinomial`` is replaced, for the duration of the run, by the
exponential race argmax(p / q) with q taken from ``synth.exp_noise`` -- after first
checking that torch's own ``multinomial(1)`` equals that race under a shared generator.
"""
import os as _os
import sys as _sys

if __name__ == "__main__" and _os.environ.get("PYTHONHASHSEED") != "1":
    # The reference's planner de-duplicates reactant sets through set() (planner/molstar.py:54), so one planner trace depends on the
    # string hash order.  Pin it before any work: re-run this script as a child under the seed the committed fixtures were made with
    # (no GPU is involved; a child process, not an exec).
    import subprocess as _sp
    _sys.exit(_sp.call([_sys.executable] + _sys.argv, env=dict(_os.environ, PYTHONHASHSEED="1")))

import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/model"
OUT = os.path.dirname(os.path.abspath(__file__))

from llamole_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------- stubs
def install_stubs():
    import torch.nn as nn

    tg = types.ModuleType("torch_geometric")
    tgu = types.ModuleType("torch_geometric.utils")
    # torch_geometric.utils used by the TRAINING forward only (diffusion_utils.py:111-124), published semantics:
    def to_dense_batch(x, batch, max_num_nodes=None):
        B = int(batch.max().item()) + 1
        counts = torch.bincount(batch, minlength=B)
        n = int(max_num_nodes if max_num_nodes is not None else counts.max())
        start = torch.cumsum(counts, 0) - counts
        pos = torch.arange(x.shape[0]) - start[batch]
        out = torch.zeros(B, n, *x.shape[1:], dtype=x.dtype)
        mask = torch.zeros(B, n, dtype=torch.bool)
        out[batch, pos] = x
        mask[batch, pos] = True
        return out, mask

    def remove_self_loops(edge_index, edge_attr=None):
        keep = edge_index[0] != edge_index[1]
        return edge_index[:, keep], (edge_attr[keep] if edge_attr is not None else None)

    def to_dense_adj(edge_index, batch, edge_attr, max_num_nodes=None):
        B = int(batch.max().item()) + 1
        counts = torch.bincount(batch, minlength=B)
        n = int(max_num_nodes if max_num_nodes is not None else counts.max())
        start = torch.cumsum(counts, 0) - counts
        g = batch[edge_index[0]]
        i, j = edge_index[0] - start[g], edge_index[1] - start[g]
        adj = torch.zeros(B, n, n, *edge_attr.shape[1:], dtype=edge_attr.dtype)
        adj.index_put_((g, i, j), edge_attr, accumulate=True)      # duplicate edges add, like PyG's scatter
        return adj

    tgu.to_dense_batch, tgu.remove_self_loops, tgu.to_dense_adj = to_dense_batch, remove_self_loops, to_dense_adj
    tgn = types.ModuleType("torch_geometric.nn")

    class MessagePassing(nn.Module):
        def __init__(self, aggr="add"):
            super().__init__()
            assert aggr == "add"

        def propagate(self, edge_index, x, edge_attr):
            msg = self.message(x_j=x[edge_index[0]], edge_attr=edge_attr)
            out = torch.zeros_like(x).index_add_(0, edge_index[1], msg)
            return self.update(out)

    def _G(batch):
        return int(batch.max().item()) + 1

    def global_add_pool(h, batch):
        return torch.zeros(_G(batch), h.shape[1], dtype=h.dtype).index_add_(0, batch, h)

    def global_max_pool(h, batch):
        out = torch.full((_G(batch), h.shape[1]), float("-inf"), dtype=h.dtype)
        return out.scatter_reduce(0, batch[:, None].expand_as(h), h, reduce="amax")

    def global_mean_pool(h, batch):
        cnt = torch.bincount(batch).clamp_min(1).to(h.dtype)[:, None]
        return global_add_pool(h, batch) / cnt

    tgn.MessagePassing = MessagePassing
    tgn.global_add_pool = global_add_pool
    tgn.global_max_pool = global_max_pool
    tgn.global_mean_pool = global_mean_pool
    tg.utils, tg.nn = tgu, tgn
    sys.modules.update({"torch_geometric": tg, "torch_geometric.utils": tgu, "torch_geometric.nn": tgn})

    rd = types.ModuleType("rdkit")
    chem = types.ModuleType("rdkit.Chem")
    allchem = types.ModuleType("rdkit.Chem.AllChem")
    rdchem = types.SimpleNamespace(BondType=types.SimpleNamespace(SINGLE=1, DOUBLE=2, TRIPLE=3, AROMATIC=4))
    chem.rdchem = rdchem
    chem.AllChem = allchem
    rd.Chem = chem
    rd.RDLogger = types.SimpleNamespace(DisableLog=lambda *a, **k: None)
    sys.modules.update({"rdkit": rd, "rdkit.Chem": chem, "rdkit.Chem.AllChem": allchem})
    rc = types.ModuleType("rdchiral")
    rcm = types.ModuleType("rdchiral.main")
    rcm.rdchiralRunText = None
    rc.main = rcm
    sys.modules.update({"rdchiral": rc, "rdchiral.main": rcm})


def check_multinomial_is_race():
    g = torch.Generator().manual_seed(123)
    p = torch.rand(4096, 16, generator=g) + 1e-3
    p = p / p.sum(-1, keepdim=True)
    g1 = torch.Generator().manual_seed(7)
    a = p.multinomial(1, generator=g1).squeeze(1)
    g2 = torch.Generator().manual_seed(7)
    q = torch.empty_like(p).exponential_(1, generator=g2)
    b = torch.argmax(p / q, dim=-1)
    assert torch.equal(a, b), "torch.multinomial(1) is not the exponential race on this build"


class NoiseFeed:
    """Replaces Tensor.multinomial by the race with queued noise tensors."""

    def __init__(self):
        self.queue = []
        self._orig = torch.Tensor.multinomial

    def push(self, *qs):
        self.queue.extend(qs)

    def __enter__(self):
        feed = self

        def multinomial(p, num_samples, replacement=False, *, generator=None):
            assert num_samples == 1 and feed.queue, "unexpected multinomial call"
            q = feed.queue.pop(0)
            assert q.shape == p.shape, (q.shape, p.shape)
            return torch.argmax(p / q, dim=-1, keepdim=True)

        torch.Tensor.multinomial = multinomial
        return self

    def __exit__(self, *a):
        torch.Tensor.multinomial = self._orig


# ----------------------------------------------------------------------------- GraphDiT goldens
from tests.cases import DIT_CASES, GIN_CASES, dit_case  # noqa: E402,F401


def gen_dit(name):
    from graph_decoder import diffusion_model as dm
    from graph_decoder import diffusion_utils as du

    cfg, meta, sd, B, seed = dit_case(name)
    N, T = meta["max_node"], cfg["diffusion_steps"]
    tmp = tempfile.mkdtemp()
    synth.write_dit_dir(tmp, cfg, meta, sd)
    model = dm.GraphDiT(os.path.join(tmp, "config.yaml"), os.path.join(tmp, "data.meta.json"), torch.float32)
    ref_sd = model.denoiser.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "state-dict key order/name mismatch vs reference"
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.init_model(tmp)
    model.eval()
    out = {}
    props, text, n_nodes = synth.make_dit_inputs(B, seed, N)
    props[0, 1] = -200.0      # exercise the no_label_index -> NaN mapping (diffusion_model.py:259)
    if B > 2:
        text[2, 5] = float("nan")   # NaN text row -> dropped embedding (conditions.py:112)
    out["props"], out["text"], out["n_nodes"] = props.numpy(), text.numpy(), n_nodes.numpy()

    # (iii) schedule tables, transition pieces
    out["betas"] = model.noise_schedule.betas.numpy()
    out["alphas_bar"] = model.noise_schedule.alphas_bar.numpy()
    out["u"] = model.transition_model.u[0].numpy()
    out["x_marg"] = model.limit_dist.X.numpy()
    out["e_marg"] = model.limit_dist.E.numpy()

    y = torch.where(props == -200.0, torch.tensor(float("nan")), props)
    mask = torch.arange(N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)

    # z_T with injected noise (sample_discrete_feature_noise)
    with NoiseFeed() as feed:
        feed.push(*synth.exp_noise(seed, T, B, N))
        zT = du.sample_discrete_feature_noise(limit_dist=model.limit_dist, node_mask=mask)
    X, E = zT.X, zT.E
    out["X_T"] = X.argmax(-1).numpy().astype(np.int8)
    # E one-hot may be all-zero (masked / diagonal): encode that as -1
    Ei = E.argmax(-1)
    Ei[E.sum(-1) == 0] = -1
    Xi = X.argmax(-1)
    Xi[X.sum(-1) == 0] = -1
    out["X_T"], out["E_T"] = Xi.numpy().astype(np.int8), Ei.numpy().astype(np.int8)

    # (i)/(ii) one denoiser call at s = T-1 (cond + uncond) incl. conditioning vectors
    s_int = T - 1
    t = (torch.full((B, 1), float(s_int)) + 1) / T
    with torch.no_grad():
        for tag, unc in (("c", False), ("u", True)):
            den = model.denoiser
            c = den.t_embedder(t) + den.y_embedder(y, False, unc) + den.txt_embedder(text, False, unc)
            out[f"cvec_{tag}"] = c.numpy()
            h = den.x_embedder(torch.cat([X, E.reshape(B, N, -1)], dim=-1))
            out[f"h0_{tag}"] = h.numpy()
            h1 = den.blocks[0](h, c, mask)
            out[f"h1_{tag}"] = h1.numpy()
            pred = den(X, E, mask, y.clone(), text, t, unconditioned=unc)
            out[f"logX_{tag}"], out[f"logE_{tag}"] = pred.X.numpy(), pred.E.numpy()

    # (iv)/(v) one full sampling step with recorded noise, plus its guided probabilities
    captured = {}
    orig_sample = du.sample_discrete_features

    def spy(probX, probE, node_mask, step=None, add_nose=True):
        captured["pX"], captured["pE"] = probX.clone(), probE.clone()
        return orig_sample(probX, probE, node_mask, step, add_nose)

    du.sample_discrete_features = spy
    try:
        with torch.no_grad(), NoiseFeed() as feed:
            feed.push(*synth.exp_noise(seed, s_int, B, N))
            s_arr = s_int * torch.ones((B, 1))
            one_hot, disc = model.sample_p_zs_given_zt(s_arr / T, (s_arr + 1) / T, X, E, y, text, mask)
    finally:
        du.sample_discrete_features = orig_sample
    out["step_pX"], out["step_pE"] = captured["pX"].numpy(), captured["pE"].numpy()
    out["step_X"], out["step_E"] = disc.X.numpy().astype(np.int8), disc.E.numpy().astype(np.int8)

    # (vi) full trajectory through the reference's own generate(), graph_to_smiles captured
    grabbed = {}
    dm.graph_to_smiles = lambda mols, dec: grabbed.setdefault("mols", mols) and [None] * len(mols)

    class FixedNodes:
        def sample_n(self, n, device):
            return n_nodes.clone()

    model.node_dist = FixedNodes()
    trace = {}
    orig_step = model.sample_p_zs_given_zt

    def traced(s, t_, X_t, E_t, yy, txt, m):
        r = orig_step(s, t_, X_t, E_t, yy, txt, m)
        si = int(round(float(s[0, 0]) * T))
        if si % 10 == 0:
            trace[si] = (r[1].X.numpy().astype(np.int8), r[1].E.numpy().astype(np.int8))
        return r

    model.sample_p_zs_given_zt = traced
    with torch.no_grad(), NoiseFeed() as feed:
        feed.push(*synth.exp_noise(seed, T, B, N))
        for s in reversed(range(T)):
            feed.push(*synth.exp_noise(seed, s, B, N))
        model.generate(props.clone(), text, -200.0)
        assert not feed.queue
    for i, (a, e) in enumerate(grabbed["mols"]):
        out[f"mol{i}_atoms"] = a.numpy().astype(np.int8)
        out[f"mol{i}_bonds"] = e.numpy().astype(np.int8)
    for si, (a, e) in trace.items():
        out[f"trace{si}_X"], out[f"trace{si}_E"] = a, e
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "ok:", {k: v.shape for k, v in list(out.items())[:6]}, "...")


def gen_dit_train(name):
    """GraphDiT.forward (training loss, diffusion_model.py:148-250, 402-438) of the reference in eval mode (no condition
    dropout / noise), with the per-graph timesteps and the forward-noise draws injected."""
    from graph_decoder import diffusion_model as dm

    cfg, meta, sd, B, seed = dit_case(name)
    N, T = meta["max_node"], cfg["diffusion_steps"]
    tmp = tempfile.mkdtemp()
    synth.write_dit_dir(tmp, cfg, meta, sd)
    model = dm.GraphDiT(os.path.join(tmp, "config.yaml"), os.path.join(tmp, "data.meta.json"), torch.float32)
    model.init_model(tmp)
    model.eval()
    x, ei, ea, batch, props, text, t_int = synth.make_dit_train_batch(meta, B, seed, T)
    captured = {}
    orig_fwd = model._forward

    def spy(noisy, txt, unconditioned=False):
        captured["X_t"], captured["E_t"] = noisy["X_t"].clone(), noisy["E_t"].clone()
        pred = orig_fwd(noisy, txt, unconditioned)
        captured["pX"], captured["pE"] = pred.X.detach().clone(), pred.E.detach().clone()
        return pred
    model._forward = spy
    orig_randint = torch.randint
    torch.randint = lambda lo, hi, size, device=None, **k: t_int.clone().view(size)
    try:
        with torch.no_grad(), NoiseFeed() as feed:
            feed.push(*synth.exp_noise(seed, T + 1, B, N))
            loss = model(x, ei, ea, batch, props.clone(), text, -200.0)
            assert not feed.queue
    finally:
        torch.randint = orig_randint
    Xi = captured["X_t"].argmax(-1)
    Xi[captured["X_t"].sum(-1) == 0] = -1
    Ei = captured["E_t"].argmax(-1)
    Ei[captured["E_t"].sum(-1) == 0] = -1
    out = {"loss": np.float32(loss.item()), "X_t": Xi.numpy().astype(np.int8), "E_t": Ei.numpy().astype(np.int8),
           "pX": captured["pX"].numpy(), "pE": captured["pE"].numpy(), "t_int": t_int.numpy()}
    np.savez_compressed(os.path.join(OUT, name + "_train.npz"), **out)
    print(name, "train ok: loss", float(loss), "t", t_int.view(-1).tolist())


# ----------------------------------------------------------------------------- GIN goldens
def gen_gin(name):
    from graph_encoder import model as enc
    from graph_predictor import model as pred

    L, H, out_dim, G, seed = GIN_CASES[name]
    x, ei, ea, batch = synth.make_mol_graphs(G, seed)
    out = {}
    sd_e = synth.make_gin_weights(L, H, "encoder", seed=seed)
    sd_p = synth.make_proj_weights(H, seed)
    clip = enc.GraphCLIP(L, H, 0.0, {"num_layer": L, "hidden_size": H, "drop_ratio": 0.0})
    assert sorted(clip.molecule_encoder.state_dict().keys()) == sorted(sd_e.keys())
    assert list(clip.molecule_projection.state_dict().keys()) == list(sd_p.keys())
    clip.molecule_encoder.load_state_dict(sd_e)
    clip.molecule_projection.load_state_dict(sd_p)
    clip.eval()
    with torch.no_grad():
        out["enc_graph"] = clip.molecule_encoder(x, ei, ea, batch).numpy()
        out["enc_out"] = clip(x, ei, ea, batch).numpy()

    sd_r = synth.make_gin_weights(L, H, "predictor", out_dim, seed)
    net = pred.GNNRetrosynthsizer(L, H, 768, 0.0, out_dim)
    assert sorted(net.state_dict().keys()) == sorted(sd_r.keys())
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(sd_r[k].shape), k
    net.load_state_dict(sd_r)
    net.eval()
    rs = np.random.RandomState(6000 + seed)
    c = torch.from_numpy(rs.standard_normal((G, 768)).astype(np.float32))
    out["c"] = c.numpy()
    with torch.no_grad():
        lg = net(x, ei, ea, batch, c)
        out["logits_c"] = lg.numpy()
        out["logits_none"] = net(x, ei, ea, batch, None).numpy()
        pr, idx = torch.topk(torch.softmax(lg, dim=1), k=50, dim=1)
        out["topk_p"], out["topk_i"] = pr.numpy(), idx.numpy().astype(np.int32)
    # host tail of sample_templates with a scripted template runner (rdchiral is absent here)
    import json
    from tests.cases import fake_template_runner
    pred.rdchiralRunText = fake_template_runner
    gp = pred.GraphPredictor(L, H, 0.0, out_dim, {"num_layer": L, "hidden_size": H, "drop_ratio": 0.0, "num_task": out_dim},
                             {i: f"T{i}" for i in range(out_dim)})
    gp.predictor.load_state_dict(sd_r)
    gp.eval()
    n0 = int((batch == 0).sum())
    e0 = (ei[0] < n0)
    import types as _t
    pg = _t.SimpleNamespace(x=x[:n0], edge_index=ei[:, e0], edge_attr=ea[e0])
    with torch.no_grad():
        r, sc, tm = gp.sample_templates(pg, c[:1], "PROD", topk=50)
    with open(os.path.join(OUT, name + "_templates.json"), "w") as f:
        json.dump({"reactants": r, "scores": sc, "templates": tm}, f)
    cm = pred.CostMLP(n_layers=1, fp_dim=2048, latent_dim=128, dropout_rate=0.1)
    cm.load_state_dict(synth.make_cost_weights(seed))
    cm.eval()
    with torch.no_grad():
        out["cost_out"] = cm(synth.make_fingerprints(4, seed)).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "ok")


# ----------------------------------------------------------------------------- planner goldens
def gen_planner():
    """Traces of the reference's molstar on scripted expansion tables (host logic, a20).  The script re-runs itself under
    PYTHONHASHSEED=1 (top of the file): the reference's set() de-duplication makes the "fail" case (a tie
    between two open nodes) depend on the string hash order; every other case gives the same trace under any seed (planner_cases.py)."""
    import json
    sys.path.insert(0, REF)
    from planner.molstar import molstar

    from tests.golden.planner_cases import CASES, make_fns

    res = {}
    for name, case in CASES.items():
        expand_fn, value_fn, log = make_fns(case)
        succ, route, iters = molstar(case["target"], 0, case["starting"], expand_fn, value_fn,
                                     iterations=case.get("iterations", 20), max_time=1e9)
        r = {"succ": bool(succ), "iters": int(iters), "expand_order": log["expand"], "value_calls": log["value"]}
        if route is not None:
            reactions, templates, costs, analysis = route.get_reaction_list()
            r.update(reactions=reactions, templates=templates, costs=[float(c) for c in costs],
                     total_cost=float(route.total_cost), length=int(route.length))
        res[name] = r
    with open(os.path.join(OUT, "planner_traces.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("planner ok", {k: (v["succ"], v["iters"]) for k, v in res.items()})


if __name__ == "__main__":
    torch.set_num_threads(8)
    install_stubs()
    check_multinomial_is_race()
    sys.path.insert(0, REF)
    which = sys.argv[1:] or ["dit", "dit_train", "gin", "planner"]
    if "dit" in which:
        for n in DIT_CASES:
            gen_dit(n)
    if "dit_train" in which:
        for n in DIT_CASES:
            gen_dit_train(n)
    if "gin" in which:
        for n in GIN_CASES:
            gen_gin(n)
    if "planner" in which:
        gen_planner()
