"""CPU oracle for the GraphDiT reverse-diffusion sampler (TEST INFRASTRUCTURE ONLY).

This file is a checker, never a product path: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The shipped path
(``llamole_amd``) never imports anything under ``oracle/`` and fails loudly when the
HIP library is missing.

It is a functional, dependency-free (torch-CPU fp32, no torch_geometric / rdkit)
restatement of the reference algorithm, deliberately kept in the reference's *dense*
formulation (one-hot float states, dense F x F transition matrices, three matmuls per
posterior) so that it pins the reference's arithmetic, not this repo's structured
re-derivation of it.  Each function cites the reference lines it follows.

Pinning: the reference has no tests / golden vectors of its own, so
this oracle is pinned against outputs of the reference itself, generated in the build
container by ``tests/golden/make_goldens.py`` (which imports the reference modules from
/root/reference with import-time stubs) and committed under ``tests/golden/*.npz``;
``tests/test_oracle_vs_golden.py`` checks every fixture.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

XDIM, EDIM, YDIM = 16, 5, 10


# ----------------------------------------------------------------------------- spec
@dataclass
class DitSpec:
    H: int
    L: int
    heads: int
    N: int
    T: int
    guide_scale: float
    x_marg: torch.Tensor        # [16]
    e_marg: torch.Tensor        # [5]
    u_xe: torch.Tensor          # [16,5]
    u_ex: torch.Tensor          # [5,16]
    u: torch.Tensor             # [F,F]
    betas: torch.Tensor         # [T+1]
    alphas_bar: torch.Tensor    # [T+1]
    node_hist: torch.Tensor     # [N+1] (unnormalised)
    active_index: torch.Tensor  # [16]

    @property
    def F(self) -> int:
        return XDIM + EDIM * self.N


def cosine_betas(T: int, s: float = 0.008) -> np.ndarray:
    """reference diffusion_utils.py:364-373 (numpy f64)."""
    steps = T + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(0.5 * np.pi * ((x / steps) + s) / (1 + s)) ** 2
    ac = ac / ac[0]
    return (1 - ac[1:] / ac[:-1]).squeeze()


def build_spec(cfg: dict, meta: dict) -> DitSpec:
    """reference diffusion_utils.py:29-59 (DataInfos), :172-187 (schedule),
    diffusion_model.py:78-103 (marginals / cross conditionals), diffusion_utils.py:273-308."""
    N = int(meta["max_node"])
    T = int(cfg["diffusion_steps"])
    atom_dist = torch.tensor(meta["atom_type_dist"], dtype=torch.float32)
    active = (atom_dist > 0).nonzero().squeeze()
    node_types = atom_dist[active]
    edge_types = torch.tensor(meta["bond_type_dist"], dtype=torch.float32)
    x_marg = node_types / node_types.sum()
    e_marg = edge_types / edge_types.sum()
    x_marg = x_marg / x_marg.sum()
    e_marg = e_marg / e_marg.sum()
    tE = torch.tensor(meta["transition_E"], dtype=torch.float32)
    xe = tE[active][:, active].sum(dim=1)            # [16,5]
    ex = xe.t()
    xe = xe / xe.sum(dim=-1, keepdim=True)
    ex = ex / ex.sum(dim=-1, keepdim=True)
    u_x = x_marg.unsqueeze(0).expand(XDIM, -1)
    u_e = e_marg.unsqueeze(0).expand(EDIM, -1)
    top = torch.cat([u_x, xe.repeat(1, N)], dim=1)
    bot = torch.cat([ex.repeat(N, 1), u_e.repeat(N, N)], dim=1)
    u = torch.cat([top, bot], dim=0).contiguous()
    betas = torch.from_numpy(cosine_betas(T)).float()
    alphas = 1 - torch.clamp(betas, min=0, max=1)
    alphas_bar = torch.exp(torch.cumsum(torch.log(alphas), dim=0))
    return DitSpec(H=int(cfg["hidden_size"]), L=int(cfg["depth"]), heads=int(cfg["num_heads"]),
                   N=N, T=T, guide_scale=float(cfg["guide_scale"]), x_marg=x_marg, e_marg=e_marg,
                   u_xe=xe, u_ex=ex, u=u, betas=betas, alphas_bar=alphas_bar,
                   node_hist=torch.tensor(meta["n_atoms_per_mol_dist"], dtype=torch.float32),
                   active_index=active)


# ----------------------------------------------------------------------------- embedders (a7)
def time_embedding(sd: Dict[str, torch.Tensor], t: torch.Tensor) -> torch.Tensor:
    """reference conditions.py:19-58; ``t`` is the *fractional* timestep in (0,1]."""
    half = 128
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t.view(-1)[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    h = F.linear(emb, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])
    return F.linear(F.silu(h), sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])


def cond_embedding(sd, y: torch.Tensor, uncond: bool) -> torch.Tensor:
    """reference conditions.py:60-98 (eval mode): per slot, NaN/unconditioned ->
    ``embedding_drop`` row, else Linear(1,H) -> softmax over H -> Linear(H,H, no bias)."""
    B = y.shape[0]
    H = sd["y_embedder.embedding_drop.weight"].shape[1]
    out = torch.zeros(B, H)
    for d in range(y.shape[1]):
        lab = y[:, d]
        drop = torch.ones_like(lab).bool() if uncond else torch.isnan(lab)
        e = torch.zeros(B, H)
        keep = ~drop
        if keep.any():
            z = F.linear(lab[keep].unsqueeze(1), sd[f"y_embedder.mlps.{d}.0.weight"],
                         sd[f"y_embedder.mlps.{d}.0.bias"])
            e[keep] = F.linear(torch.softmax(z, dim=1), sd[f"y_embedder.mlps.{d}.2.weight"])
        e[drop] += sd["y_embedder.embedding_drop.weight"][d]
        out = out + e
    return out


def text_embedding(sd, txt: torch.Tensor, uncond: bool) -> torch.Tensor:
    """reference conditions.py:100-123 (eval mode)."""
    B = txt.shape[0]
    drop = torch.ones(B).bool() if uncond else torch.isnan(txt.sum(dim=1))
    H = sd["txt_embedder.linear.weight"].shape[0]
    e = torch.zeros(B, H)
    keep = ~drop
    if keep.any():
        e[keep] = F.linear(txt[keep], sd["txt_embedder.linear.weight"], sd["txt_embedder.linear.bias"])
    e[drop] += sd["txt_embedder.embedding_drop.weight"][0]
    return e


def conditioning(sd, y, txt, t, uncond: bool) -> torch.Tensor:
    """c = c_t + c_y + c_txt (reference transformer.py:98-101)."""
    return time_embedding(sd, t) + cond_embedding(sd, y, uncond) + text_embedding(sd, txt, uncond)


# ----------------------------------------------------------------------------- denoiser (a6, a8, a9, a10)
def _ln(x, w=None, b=None):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def attention(sd, p: str, x: torch.Tensor, mask: torch.Tensor, heads: int) -> torch.Tensor:
    """reference layers.py:56-87: per-head LayerNorm on q,k; key mask valid_i & valid_j with
    padded query rows opened to every key; softmax(q k^T / sqrt(hd)) v; output projection."""
    B, N, D = x.shape
    hd = D // heads
    qkv = F.linear(x, sd[p + "qkv.weight"]).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = _ln(q, sd[p + "q_norm.weight"], sd[p + "q_norm.bias"])
    k = _ln(k, sd[p + "k_norm.weight"], sd[p + "k_norm.bias"])
    allow = (mask[:, None, :, None] & mask[:, None, None, :]).expand(-1, heads, N, N).clone()
    allow[allow.sum(dim=-1) == 0] = True
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    s = s.masked_fill(~allow, float("-inf"))
    o = torch.softmax(s, dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, N, D)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def block(sd, i: int, x, c, mask, heads: int):
    """reference transformer.py:132-145 (post-norm AdaLN, Softsign on the modulation)."""
    p = f"blocks.{i}."
    m = F.linear(c, sd[p + "adaLN_modulation.0.weight"], sd[p + "adaLN_modulation.0.bias"])
    m = F.softsign(F.linear(F.silu(m), sd[p + "adaLN_modulation.2.weight"], sd[p + "adaLN_modulation.2.bias"]))
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = m.chunk(6, dim=1)
    a = _ln(attention(sd, p + "attn.", x, mask, heads))
    x = x + g_a.unsqueeze(1) * (a * (1 + sc_a.unsqueeze(1)) + sh_a.unsqueeze(1))
    h = F.linear(F.gelu(F.linear(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])),
                 sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    h = _ln(h)
    return x + g_m.unsqueeze(1) * (h * (1 + sc_m.unsqueeze(1)) + sh_m.unsqueeze(1))


def output_layer(sd, x, X_in, E_in, c, mask):
    """reference transformer.py:163-187 followed by PlaceHolder.mask (diffusion_utils.py:93-108)."""
    B, N, _ = X_in.shape
    p = "output_layer."
    h = F.linear(F.gelu(F.linear(x, sd[p + "xedecoder.fc1.weight"], sd[p + "xedecoder.fc1.bias"])),
                 sd[p + "xedecoder.fc2.weight"], sd[p + "xedecoder.fc2.bias"])
    m = F.linear(F.silu(F.linear(c, sd[p + "adaLN_modulation.0.weight"], sd[p + "adaLN_modulation.0.bias"])),
                 sd[p + "adaLN_modulation.2.weight"], sd[p + "adaLN_modulation.2.bias"])
    shift, scale = m.chunk(2, dim=1)
    h = _ln(h) * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)
    atom = X_in + h[:, :, :XDIM]
    bond = E_in + h[:, :, XDIM:].reshape(B, N, N, EDIM)
    both_invalid = (~mask)[:, :, None] & (~mask)[:, None, :]
    eye = torch.eye(N, dtype=torch.bool).unsqueeze(0).expand(B, -1, -1)
    bond = bond.masked_fill(both_invalid[..., None], 0).masked_fill(eye[..., None], 0)
    bond = 0.5 * (bond + bond.transpose(1, 2))
    xm = mask.unsqueeze(-1)
    atom = atom * xm
    bond = bond * xm.unsqueeze(2) * xm.unsqueeze(1)
    return atom, bond


def denoiser(sd, spec: DitSpec, X, E, mask, y, txt, t, uncond: bool, return_hidden: bool = False):
    """reference transformer.py:93-108.  X [B,N,16], E [B,N,N,5] one-hot floats."""
    B, N, _ = X.shape
    h = torch.cat([X, E.reshape(B, N, -1)], dim=-1)
    h = _ln(F.linear(h, sd["x_embedder.0.weight"]), sd["x_embedder.1.weight"], sd["x_embedder.1.bias"])
    c = conditioning(sd, y, txt, t, uncond)
    hs = [h]
    for i in range(spec.L):
        h = block(sd, i, h, c, mask, spec.heads)
        hs.append(h)
    lx, le = output_layer(sd, h, X, E, c, mask)
    if return_hidden:
        return lx, le, c, hs
    return lx, le


# ----------------------------------------------------------------------------- posterior (a5, a12, a13)
def posterior(spec: DitSpec, logX, logE, X_t, E_t, s_int: int):
    """reference diffusion_model.py:328-364 (get_prob) + diffusion_utils.py:316-349, 476-492.
    Dense: builds Qt, Qsb, Qtb [F,F] and does the three matmuls."""
    B, N, _ = X_t.shape
    t_int = s_int + 1
    beta_t = spec.betas[t_int]
    ab_s = spec.alphas_bar[s_int]
    ab_t = spec.alphas_bar[t_int]
    Fd = spec.F
    eye = torch.eye(Fd)
    Qt = beta_t * spec.u + (1 - beta_t) * eye
    Qsb = ab_s * eye + (1 - ab_s) * spec.u
    Qtb = ab_t * eye + (1 - ab_t) * spec.u
    pX = torch.softmax(logX, dim=-1)
    pE = torch.softmax(logE, dim=-1)
    Xt_all = torch.cat([X_t, E_t.reshape(B, N, -1)], dim=-1)
    p_all = torch.cat([pX, pE.reshape(B, N, -1)], dim=-1)
    left = Xt_all @ Qt.t()
    right = p_all @ Qsb
    den = (Qtb @ Xt_all.transpose(-1, -2)).transpose(-1, -2)
    un = left * right / den.clamp_min(1e-5)
    unX = un[:, :, :XDIM].clone()
    unE = un[:, :, XDIM:].reshape(B, N * N, EDIM).clone()
    unX[unX.sum(dim=-1) == 0] = 1e-5
    unE[unE.sum(dim=-1) == 0] = 1e-5
    prX = unX / unX.sum(dim=-1, keepdim=True)
    prE = (unE / unE.sum(dim=-1, keepdim=True)).reshape(B, N, N, EDIM)
    return prX, prE


def guided_probs(sd, spec: DitSpec, X_t, E_t, mask, y, txt, s_int: int, return_logits: bool = False):
    """reference diffusion_model.py:309-382: cond + uncond passes, CFG combine, renormalise.
    ``return_logits``: also return the denoiser outputs ``(lx, le, ux, ue)`` behind the probabilities (ux / ue are None
    without guidance) -- same arithmetic, for tests that compare logits and probabilities of one step."""
    B = X_t.shape[0]
    t = torch.full((B, 1), float(s_int), dtype=torch.float32)
    t = (t + 1) / spec.T
    lx, le = denoiser(sd, spec, X_t, E_t, mask, y, txt, t, uncond=False)
    pX, pE = posterior(spec, lx, le, X_t, E_t, s_int)
    ux = ue = None
    if spec.guide_scale is not None and spec.guide_scale != 1:
        ux, ue = denoiser(sd, spec, X_t, E_t, mask, y, txt, t, uncond=True)
        uX, uE = posterior(spec, ux, ue, X_t, E_t, s_int)
        pX = uX * (pX / uX.clamp_min(1e-5)) ** spec.guide_scale
        pE = uE * (pE / uE.clamp_min(1e-5)) ** spec.guide_scale
        pX = pX / pX.sum(dim=-1, keepdim=True).clamp_min(1e-5)
        pE = pE / pE.sum(dim=-1, keepdim=True).clamp_min(1e-5)
    if return_logits:
        return pX, pE, (lx, le, ux, ue)
    return pX, pE


# ----------------------------------------------------------------------------- sampling (a14)
def race(p: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """``p.multinomial(1)`` restated as the exponential race torch runs on CPU:
    argmax(p / q) with q ~ Exp(1) of p's shape (checked against torch.multinomial under a
    shared generator in tests/golden/make_goldens.py)."""
    return torch.argmax(p / q, dim=-1)


def sample_features(pX, pE, mask, qx, qe):
    """reference diffusion_utils.py:376-413.  Noise contract: qx [B*N,16] then qe [B*N*N,5]."""
    B, N, _ = pX.shape
    pX = pX.clone()
    pE = pE.clone()
    pX[~mask] = 1 / XDIM
    pX = pX.reshape(B * N, -1).clamp_min(1e-5)
    pX = pX / pX.sum(dim=-1, keepdim=True)
    Xs = race(pX, qx).reshape(B, N)
    inv = ~(mask.unsqueeze(1) * mask.unsqueeze(2))
    eye = torch.eye(N, dtype=torch.bool).unsqueeze(0).expand(B, -1, -1)
    pE[inv] = 1 / EDIM
    pE[eye] = 1 / EDIM
    pE = pE.reshape(B * N * N, -1).clamp_min(1e-5)
    pE = pE / pE.sum(dim=-1, keepdim=True)
    Es = race(pE, qe).reshape(B, N, N)
    Es = torch.triu(Es, diagonal=1)
    Es = Es + Es.transpose(1, 2)
    return Xs, Es


def to_onehot_masked(Xs, Es, mask):
    """reference diffusion_model.py:388-399 (one_hot + PlaceHolder.mask)."""
    X = F.one_hot(Xs, XDIM).float() * mask.unsqueeze(-1)
    xm = mask.unsqueeze(-1)
    E = F.one_hot(Es, EDIM).float() * xm.unsqueeze(2) * xm.unsqueeze(1)
    return X, E


def initial_state(spec: DitSpec, mask, qx, qe):
    """reference diffusion_utils.py:495-518: sample the limit marginals, keep the strict upper
    triangle, symmetrise, mask.  (The diagonal of E is all-zero here, unlike later steps
    where it is the one-hot of class 0.)"""
    B, N = mask.shape
    Xs = race(spec.x_marg[None, :].expand(B * N, -1), qx).reshape(B, N)
    Es = race(spec.e_marg[None, :].expand(B * N * N, -1), qe).reshape(B, N, N)
    X = F.one_hot(Xs, XDIM).float()
    E = F.one_hot(Es, EDIM).float()
    tri = torch.triu(torch.ones(N, N), diagonal=1)[None, :, :, None]
    E = E * tri
    E = E + E.transpose(1, 2)
    xm = mask.unsqueeze(-1)
    return X * xm, E * xm.unsqueeze(2) * xm.unsqueeze(1)


def collapse(X, E, mask):
    """reference diffusion_utils.py:98-103 (collapse=True)."""
    Xi = X.argmax(-1)
    Ei = E.argmax(-1)
    Xi[~mask] = -1
    Ei[~(mask.unsqueeze(1) & mask.unsqueeze(2))] = -1
    return Xi, Ei


def sample_n_nodes(spec: DitSpec, batch: int, generator=None) -> torch.Tensor:
    """reference diffusion_utils.py:143-170 (Categorical(hist).sample)."""
    p = spec.node_hist / spec.node_hist.sum()
    return torch.multinomial(p, batch, replacement=True, generator=generator)


def generate(sd, spec: DitSpec, y, txt, n_nodes, noise_fn: Callable[[int], Tuple[torch.Tensor, torch.Tensor]],
             no_label_index: float = -200.0, trace_every: int = 0, step_hook=None):
    """reference diffusion_model.py:252-304 up to the integer graphs (the rdkit tail,
    molecule_utils.py:49-111, is host chemistry and not part of the oracle).
    ``noise_fn(step)`` returns (qx, qe); step == T is the initial z_T draw, then T-1 .. 0.
    ``step_hook(s_int, X_t, E_t, pX, pE, logits)`` (tests): sees every step's input state, guided probabilities and the
    denoiser outputs behind them."""
    y = torch.where(y == no_label_index, torch.tensor(float("nan")), y)
    B = y.shape[0]
    mask = torch.arange(spec.N).unsqueeze(0).expand(B, -1) < n_nodes.unsqueeze(1)
    X, E = initial_state(spec, mask, *noise_fn(spec.T))
    trace = {}
    for s_int in reversed(range(spec.T)):
        if step_hook is not None:
            pX, pE, logits = guided_probs(sd, spec, X, E, mask, y, txt, s_int, return_logits=True)
            step_hook(s_int, X, E, pX, pE, logits)
        else:
            pX, pE = guided_probs(sd, spec, X, E, mask, y, txt, s_int)
        Xs, Es = sample_features(pX, pE, mask, *noise_fn(s_int))
        X, E = to_onehot_masked(Xs, Es, mask)
        if trace_every and s_int % trace_every == 0:
            trace[s_int] = collapse(X.clone(), E.clone(), mask)
    Xi, Ei = collapse(X, E, mask)
    mols = [(Xi[i, :n_nodes[i]].clone(), Ei[i, :n_nodes[i], :n_nodes[i]].clone()) for i in range(B)]
    return mols, (Xi, Ei), trace


# ----------------------------------------------------------------------------- training forward (a22 / f4)
def to_dense(spec: DitSpec, x, edge_index, edge_attr, batch):
    """reference diffusion_model.py:152-160 + diffusion_utils.py:111-139 (to_dense, encode_no_edge) with the published
    semantics of torch_geometric's to_dense_batch / remove_self_loops / to_dense_adj.  Returns the UNMASKED one-hot
    X [B,N,16], E [B,N,N,5] (no-bond class set on every non-diagonal pair, padding included) and node_mask [B,N]."""
    N = spec.N
    data_x = F.one_hot(x, num_classes=118).float()[:, spec.active_index]
    data_e = F.one_hot(edge_attr, num_classes=EDIM).float()
    B = int(batch.max().item()) + 1
    counts = torch.bincount(batch, minlength=B)
    start = torch.cumsum(counts, 0) - counts
    pos = torch.arange(x.shape[0]) - start[batch]
    X = torch.zeros(B, N, XDIM)
    mask = torch.zeros(B, N, dtype=torch.bool)
    X[batch, pos] = data_x
    mask[batch, pos] = True
    keep = edge_index[0] != edge_index[1]
    ei, de = edge_index[:, keep], data_e[keep]
    g = batch[ei[0]]
    E = torch.zeros(B, N, N, EDIM)
    E.index_put_((g, ei[0] - start[g], ei[1] - start[g]), de, accumulate=True)
    no_edge = E.sum(dim=3) == 0
    E[..., 0][no_edge] = 1
    E[torch.eye(N, dtype=torch.bool).unsqueeze(0).expand(B, -1, -1)] = 0
    return X, E, mask


def apply_noise(spec: DitSpec, X, E, mask, t_int, qx, qe):
    """reference diffusion_model.py:197-250: z_t ~ q(z_t | z_0) through Q_bar_t = a_bar_t I + (1 - a_bar_t) u
    (diffusion_utils.py:332-349), sampled with the race noise qx [B*N,16], qe [B*N*N,5]; masked one-hot output."""
    B, N, _ = X.shape
    ab = spec.alphas_bar[t_int.view(-1).long()].view(B, 1, 1)
    Fd = spec.F
    Qtb = ab * torch.eye(Fd).unsqueeze(0) + (1 - ab) * spec.u.unsqueeze(0)
    prob = torch.cat([X, E.reshape(B, N, -1)], dim=-1) @ Qtb
    pX, pE = prob[:, :, :XDIM], prob[:, :, XDIM:].reshape(B, N, N, EDIM)
    Xs, Es = sample_features(pX, pE, mask, qx, qe)
    return to_onehot_masked(Xs, Es, mask)


def train_loss(spec: DitSpec, pred_X, pred_E, true_X, true_E, lambda_train=(1, 10)):
    """reference TrainLossDiscrete.forward, diffusion_model.py:402-438."""
    tX, tE = true_X.reshape(-1, XDIM), true_E.reshape(-1, EDIM)
    pX, pE = pred_X.reshape(-1, XDIM), pred_E.reshape(-1, EDIM)
    mX, mE = (tX != 0).any(dim=-1), (tE != 0).any(dim=-1)
    lx = F.cross_entropy(pX[mX], tX[mX].argmax(dim=-1), reduction="mean")
    le = F.cross_entropy(pE[mE], tE[mE].argmax(dim=-1), reduction="mean")
    return lambda_train[0] * lx + lambda_train[1] * le


def train_forward(sd, spec: DitSpec, x, edge_index, edge_attr, batch, props, text, no_label_index, t_int, qx, qe,
                  lambda_train=(1, 10)):
    """reference GraphDiT.forward (diffusion_model.py:148-173) in eval mode: no condition dropout / noise
    (conditions.py:84-94 only act when ``train``); t_int [B,1] in 0..T is what torch.randint draws there."""
    y = torch.where(props == no_label_index, torch.tensor(float("nan")), props)
    X, E, mask = to_dense(spec, x, edge_index, edge_attr, batch)
    X_t, E_t = apply_noise(spec, X, E, mask, t_int, qx, qe)
    t = t_int.float() / spec.T
    lx, le = denoiser(sd, spec, X_t, E_t, mask, y, text, t, uncond=False)
    return train_loss(spec, lx, le, X, E, lambda_train), (X_t, E_t, lx, le)
