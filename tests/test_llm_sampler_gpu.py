"""ll_sample_token_bf16 (fused temperature / top-p / multinomial sampler + decode-loop bookkeeping) through the C ABI
against PyTorch: argmax, nucleus boundary, sampled distribution, NaN/inf sanitisation, stop/pad/position bookkeeping."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _key_to_value(key: int) -> float:
    x = (key & 0x7fff) if (key & 0x8000) else (~key & 0xffff)
    return float(torch.tensor([x], dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float()) if x < 0x8000 else \
        float(torch.tensor([x - 0x10000], dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float())


class Sampler:
    def __init__(self, B, V, max_new=64, eos=(), pad=0, tap=True, split=False):
        from llamole_amd import _lib
        self.tap = tap          # False: no dbg tap, the way the decode loop calls it (top-k then counts only what can survive it)
        self.split = split      # True: with a workspace (ll_sample_token_topk_ws_bf16): candidates launch + finish launch under top-k
        self.ws = None
        self.lib, self._lib = _lib.load(), _lib
        d = "cuda"
        self.B, self.V = B, V
        self.seed = torch.zeros(1, dtype=torch.long, device=d)
        self.eos = torch.full((8,), -1, dtype=torch.long, device=d)
        if eos:
            self.eos[:len(eos)] = torch.tensor(eos, device=d)
        self.pad = pad
        self.done = torch.zeros(B, dtype=torch.uint8, device=d)
        self.tok = torch.zeros(B, dtype=torch.long, device=d)
        self.out = torch.full((B, max_new), -7, dtype=torch.long, device=d)
        self.step = torch.zeros(B, dtype=torch.long, device=d)
        self.posid = torch.zeros(B, dtype=torch.long, device=d)
        self.pos = torch.zeros(1, dtype=torch.long, device=d)
        self.dbg = torch.zeros(B, 4, dtype=torch.int64, device=d)

    def __call__(self, logits, temperature=1.0, top_p=1.0, greedy=False, advance=1, top_k=None):
        inv = float(np.float32(1.0) / np.float32(temperature))
        tail = (self.seed.data_ptr(), self.eos.data_ptr(), 8, self.pad, self.done.data_ptr(),
                self.tok.data_ptr(), self.out.data_ptr(), self.out.stride(0), self.out.shape[1],
                self.step.data_ptr(), self.posid.data_ptr(), self.pos.data_ptr(), advance,
                self.dbg.data_ptr() if self.tap else None, torch.cuda.current_stream().cuda_stream)
        if self.split:
            if self.ws is None:
                self.ws = torch.zeros(int(self.lib.ll_sample_workspace_bytes(self.B)), dtype=torch.uint8, device="cuda")
            rc = self.lib.ll_sample_token_topk_ws_bf16(logits.data_ptr(), logits.stride(0), self.B, self.V, inv, top_p, int(top_k or 0), int(greedy),
                                                       *tail[:-1], self.ws.data_ptr(), self.ws.numel(), tail[-1])
        elif top_k is None:
            rc = self.lib.ll_sample_token_bf16(logits.data_ptr(), logits.stride(0), self.B, self.V, inv, top_p, int(greedy), *tail)
        else:
            rc = self.lib.ll_sample_token_topk_bf16(logits.data_ptr(), logits.stride(0), self.B, self.V, inv, top_p, int(top_k), int(greedy), *tail)
        self._lib.check(rc, "ll_sample_token_bf16")
        return self.tok.clone()


def _rows_for_bound_test(V, g):
    """Eight rows that stress the top-k lower bound: smooth, peaked, heavily tied, constant, inf / NaN, a handful of tokens far above a
    floor that lies outside the key window, everything negative, one dominant token."""
    r = torch.randn(8, V, generator=g)
    rows = [r[0] * 2.5, r[1] * 8.0, (r[2] * 2).round() / 2, torch.full((V,), 1.5), r[4] * 3.0, torch.full((V,), -1.0e30), -r[6].abs() * 5 - 3, r[7] * 0.3]
    rows[4][torch.randint(0, V, (40,), generator=g)] = float("inf")
    rows[4][torch.randint(0, V, (40,), generator=g)] = float("nan")
    rows[4][torch.randint(0, V, (40,), generator=g)] = float("-inf")
    rows[5][torch.randint(0, V, (10,), generator=g)] = 1000.0
    rows[5][torch.randint(0, V, (10,), generator=g)] = 996.0
    rows[7][V // 3] = 40.0
    return torch.stack(rows).bfloat16().cuda()


@pytest.mark.parametrize("V", [2048, 32000, 128256, 152064, 163840])
def test_top_k_shortcuts_leave_every_token_unchanged(V):
    """Two shortcuts under top-k, both against the whole-row count that the tests below pin to the HF warpers (same seeds, same rows,
    every token equal): (a) without the dbg tap the one-workgroup sampler counts only keys that can survive top-k (a lower bound from the
    per-thread maxima, one packed compare per pair of keys, a histogram scan over the few bins above the bound); (b) with a workspace and
    top_k <= 128 the work is split -- V/2048 workgroups per row hand on candidates, one workgroup per row finishes on them; rows whose
    candidates overflow the list (the constant row: every token ties) or leave the key window fall back on the device."""
    g = torch.Generator().manual_seed(V + 1)
    rows = _rows_for_bound_test(V, g)
    for top_k, top_p, temperature in ((50, 0.9, 0.7), (1, 1.0, 1.0), (5, 0.5, 1.3), (128, 0.95, 0.6), (300, 0.95, 0.6), (1024, 1.0, 1.0), (2000, 0.9, 1.0),
                                      (0, 0.9, 0.7)):
        for lo in (0, 4):                        # the split takes up to four rows per call: the eight stress rows in two halves
            logits = rows[lo:lo + 4].contiguous()
            a, b, c = Sampler(4, V, max_new=48), Sampler(4, V, max_new=48, tap=False), Sampler(4, V, max_new=48, tap=False, split=True)
            for s in (a, b, c):
                s.seed.fill_(1234 + top_k)
            for _ in range(48):
                for s in (a, b, c):
                    s(logits, temperature=temperature, top_p=top_p, top_k=top_k)
            assert torch.equal(a.out, b.out), (V, top_k)
            assert torch.equal(a.out, c.out), (V, top_k, (a.out != c.out).any(dim=1).tolist())
            assert torch.equal(a.step, c.step) and torch.equal(a.tok, c.tok) and torch.equal(a.posid, c.posid) and torch.equal(a.pos, c.pos)
            assert int(c.ws[:4 * 16].view(torch.int32).abs().sum()) == 0                          # every row's list header is back at zero
            if top_k == 50 and lo == 0:
                assert len(set(a.out[0].tolist())) > 8 and len(set(a.out[3].tolist())) > 8     # real draws, not a constant
    big = Sampler(8, V, max_new=8, tap=False, split=True)         # more than four rows with a workspace: one launch, same tokens
    ref = Sampler(8, V, max_new=8, tap=False)
    for _ in range(8):
        big(rows, temperature=0.7, top_p=0.9, top_k=50)
        ref(rows, temperature=0.7, top_p=0.9, top_k=50)
    assert torch.equal(big.out, ref.out) and int(big.ws[:8 * 16].view(torch.int32).abs().sum()) == 0


def test_split_sampler_greedy_and_single_row():
    """The workspace changes nothing for greedy decoding (one launch as before) and works at one row (the decode loop's shape)."""
    V = 152064
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(1, V, generator=g) * 3).bfloat16().cuda()
    a, c = Sampler(1, V, max_new=32, tap=False), Sampler(1, V, max_new=32, tap=False, split=True)
    for _ in range(16):
        a(logits, greedy=True, top_k=50)
        c(logits, greedy=True, top_k=50)
    for _ in range(16):
        a(logits, temperature=0.6, top_p=0.9, top_k=50)
        c(logits, temperature=0.6, top_p=0.9, top_k=50)
    assert torch.equal(a.out, c.out) and a.out[0, 0].item() == int(logits.float().argmax())
    assert len(set(a.out[0, 16:].tolist())) > 1


@pytest.mark.parametrize("V", [2048, 32000, 128256, 152064])
def test_greedy_is_argmax_lowest_index(V):
    g = torch.Generator().manual_seed(V)
    logits = torch.randn(3, V, generator=g).bfloat16()
    mx = logits.float().max(dim=1).values
    for b, idxs in enumerate([(5, V - 3), (V // 2, V // 2 + 1), (V - 1,)]):   # ties on the maximum
        for i in idxs:
            logits[b, i] = mx[b] + 1
    logits = logits.cuda()
    s = Sampler(3, V)
    tok = s(logits, greedy=True)
    assert tok.tolist() == [5, V // 2, V - 1]
    assert tok.tolist() == logits.float().argmax(dim=1).tolist()
    assert s.out[:, 0].tolist() == tok.tolist() and s.step.tolist() == [1, 1, 1]
    assert s.posid.tolist() == [1, 1, 1] and s.pos.item() == 1


@pytest.mark.parametrize("V,scale,temperature,top_p", [(2048, 3.0, 0.6, 0.9), (152064, 0.5, 0.6, 0.9), (152064, 4.0, 1.0, 0.5),
                                                       (128256, 2.0, 0.7, 0.95), (32000, 1.0, 1.3, 1.0)])
def test_nucleus_boundary_matches_reference(V, scale, temperature, top_p):
    """Boundary value and kept mass against an f64 PyTorch evaluation of the same rule, and against the HF warper's kept
    set (ascending cumulative sum > 1 - top_p): HF's set is contained in ours, the extras are tied with the boundary."""
    g = torch.Generator().manual_seed(int(V + 100 * scale))
    B = 2
    logits = (torch.randn(B, V, generator=g) * scale).bfloat16().cuda()
    s = Sampler(B, V)
    s.seed.fill_(1234)
    tok = s(logits, temperature=temperature, top_p=top_p)
    dbg = s.dbg.cpu()
    inv = float(np.float32(1.0) / np.float32(temperature))
    for b in range(B):
        sc = (logits[b].float() * inv).double().cpu()
        p = torch.softmax(sc, dim=0)
        vals, inverse = torch.unique(sc, return_inverse=True)           # ascending distinct values
        mass = torch.zeros_like(vals).scatter_add_(0, inverse, p)
        above = mass.flip(0).cumsum(0).flip(0) - mass                   # mass strictly above each distinct value
        kept_vals = vals[above < top_p] if top_p < 1.0 else vals
        tau_ref = kept_vals.min().item()
        if top_p < 1.0:
            tau = _key_to_value(int(dbg[b, 2])) * inv
            assert abs(tau - tau_ref) <= 1e-6 * max(1.0, abs(tau_ref)), (tau, tau_ref)
        else:
            assert int(dbg[b, 0]) == int(dbg[b, 1])                     # no filtering: kept mass is all of Z
        kept = sc >= tau_ref - 1e-9
        Z, M = float(dbg[b, 0]), float(dbg[b, 1])
        assert abs(M / Z - p[kept].sum().item()) < 2e-5
        assert bool(kept[tok[b]])                                       # the sampled token is in the nucleus
        assert abs(_key_to_value(int(dbg[b, 3])) - logits[b, tok[b]].float().item()) == 0.0
        if top_p < 1.0:                                                 # HF TopPLogitsWarper on the same scores
            srt, idx = torch.sort(sc, descending=False)
            cum = torch.softmax(srt, dim=0).cumsum(0)
            remove = cum <= (1 - top_p)
            remove[-1] = False
            kept_hf = torch.ones(V, dtype=torch.bool)
            kept_hf[idx[remove]] = False
            assert bool((kept | ~kept_hf).all())                        # HF's nucleus is a subset of ours
            extra = kept & ~kept_hf
            assert bool((sc[extra] <= tau_ref + 1e-9).all())            # extras are ties of the boundary value


@pytest.mark.parametrize("V", [2048, 152064])
@pytest.mark.parametrize("top_k,top_p,temperature", [(50, 0.9, 0.6), (50, 1.0, 1.0), (1, 0.9, 0.6), (7, 0.5, 1.3), (5000, 0.95, 0.8)])
def test_top_k_then_top_p_matches_hf_warpers(V, top_k, top_p, temperature):
    """The reference samples with top_k = 50 ahead of top_p (GeneratingArguments default).  Against transformers' own
    TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper on the same scores: the kernel's kept set contains HF's,
    extras are ties of HF's lowest kept value; the kept mass over the top-k mass matches; the drawn token is in the set."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(V + top_k)
    B = 2
    logits = (torch.randn(B, V, generator=g) * 3.0).bfloat16().cuda()
    logits[0, 10:14] = logits[0].max()                                  # exact ties at the top
    s = Sampler(B, V)
    s.seed.fill_(77)
    tok = s(logits, temperature=temperature, top_p=top_p, top_k=top_k)
    dbg = s.dbg.cpu()
    inv = float(np.float32(1.0) / np.float32(temperature))
    sc = (logits.float() * inv).cpu()
    hf = TopKLogitsWarper(top_k)(None, TemperatureLogitsWarper(temperature)(None, logits.float().cpu()))
    if top_p < 1.0:
        hf = TopPLogitsWarper(top_p)(None, hf)
    kept_hf = torch.isfinite(hf)
    for b in range(B):
        tau = _key_to_value(int(dbg[b, 2])) * inv
        kept = sc[b] >= tau - 1e-6 * max(1.0, abs(tau))
        assert bool((kept | ~kept_hf[b]).all())                         # HF's set is a subset of ours
        extra = kept & ~kept_hf[b]
        assert bool((sc[b][extra] <= sc[b][kept_hf[b]].min() + 1e-6).all())      # extras tie with HF's lowest kept value
        assert int(kept.sum()) >= 1 and bool(kept[tok[b]])
        # kept mass relative to the top-k mass (f64)
        p = torch.softmax(sc[b].double(), dim=0)
        kth = torch.topk(sc[b], min(top_k, V)).values[-1]
        assert abs(float(dbg[b, 1]) / float(dbg[b, 0]) - (p[kept].sum() / 1.0).item()) < 3e-5
        assert int((sc[b] >= kth).sum()) >= int(kept.sum()) or top_p >= 1.0


def test_sampled_distribution_matches_softmax():
    """8000 draws from a peaked 2048-token row: chi-square of the head tokens against softmax of the nucleus."""
    V, B, steps = 2048, 4, 2000
    g = torch.Generator().manual_seed(3)
    base = (torch.randn(V, generator=g) * 2.5).bfloat16()
    logits = base.repeat(B, 1).cuda()
    s = Sampler(B, V, max_new=steps)
    s.seed.fill_(99)
    for _ in range(steps):
        s(logits, temperature=0.8, top_p=0.9)
    draws = s.out.flatten().cpu()
    assert (draws >= 0).all()
    inv = float(np.float32(1.0) / np.float32(0.8))
    sc = (base.float() * inv).double()
    p = torch.softmax(sc, dim=0)
    vals, inverse = torch.unique(sc, return_inverse=True)
    mass = torch.zeros_like(vals).scatter_add_(0, inverse, p)
    above = mass.flip(0).cumsum(0).flip(0) - mass
    tau = vals[above < 0.9].min()
    kept = sc >= tau
    q = torch.where(kept, p, torch.zeros_like(p))
    q = q / q.sum()
    counts = torch.bincount(draws, minlength=V).double()
    assert counts[~kept].sum() == 0                                    # nothing outside the nucleus is ever drawn
    n = counts.sum()
    big = q * n >= 20
    chi2 = (((counts[big] - q[big] * n) ** 2) / (q[big] * n)).sum().item() + \
        ((counts[~big & kept].sum() - q[~big & kept].sum() * n) ** 2 / max(q[~big & kept].sum().item() * n, 1e-9)).item()
    dof = int(big.sum())
    assert chi2 < dof + 6 * (2 * dof) ** 0.5, (chi2, dof)
    # reproducible: same seed, same steps -> same stream; another seed -> a different one
    s2 = Sampler(B, V, max_new=steps)
    s2.seed.fill_(99)
    for _ in range(50):
        s2(logits, temperature=0.8, top_p=0.9)
    assert torch.equal(s2.out[:, :50], s.out[:, :50])
    s3 = Sampler(B, V, max_new=steps)
    s3.seed.fill_(100)
    for _ in range(50):
        s3(logits, temperature=0.8, top_p=0.9)
    assert not torch.equal(s3.out[:, :50], s.out[:, :50])


def test_sanitisation_and_bookkeeping():
    V = 2048
    logits = torch.zeros(3, V).bfloat16()
    logits[0, 7] = float("nan")          # NaN counts as 0.0 (torch.nan_to_num in the HF path)
    logits[0, 11] = float("inf")         # +inf wins
    logits[1, :] = float("-inf")
    logits[1, 100] = -3.0                # everything else is -inf -> only token 100 has mass
    logits[2, 5] = 30.0
    logits = logits.cuda()
    s = Sampler(3, V, eos=(5, 9), pad=1)
    tok = s(logits, temperature=0.6, top_p=0.9, advance=0)
    assert tok.tolist() == [11, 100, 5]
    assert s.done.tolist() == [0, 0, 1]                  # row 2 sampled an EOS id
    assert s.posid.tolist() == [0, 0, 0] and s.pos.item() == 0 and s.step.tolist() == [1, 1, 1]
    tok = s(logits, temperature=0.6, top_p=0.9, advance=1)
    assert tok.tolist() == [11, 100, 1]                  # a stopped row emits pad from now on
    assert s.out[:, :3].tolist() == [[11, 11, -7], [100, 100, -7], [5, 1, -7]]
    assert s.posid.tolist() == [1, 1, 1] and s.pos.item() == 1
    # C-ABI error behaviour
    lib = s.lib
    assert lib.ll_sample_token_bf16(logits.data_ptr(), V, 3, 2047, 1.0, 0.9, 0, s.seed.data_ptr(), s.eos.data_ptr(), 8, 0,
                                    s.done.data_ptr(), s.tok.data_ptr(), s.out.data_ptr(), 64, 64, s.step.data_ptr(), None,
                                    None, 0, None, None) == -1
    assert b"multiple of 8" in lib.ll_last_error()
