"""Stress of the sampler's top-k shortcuts: random vocabularies, row counts, distributions (smooth, peaked, tied, masked to -inf, NaN / inf,
constant) and sampling settings -- every token of the tap-less one-workgroup path and of the split path must equal the whole-row count's.
python tools/sampler_stress.py [cases] [seed]"""
import os
import sys

os.environ.setdefault("LLAMOLE_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from test_llm_sampler_gpu import Sampler  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))      # noqa: E731
rf = lambda lo, hi: float(torch.rand(1, generator=g)) * (hi - lo) + lo     # noqa: E731
bad = 0
for case in range(cases):
    V = [2048, 4096, 8200, 32000, 50264, 128256, 151936, 152064, 163840][ri(0, 8)]
    B = ri(1, 4)
    kind = ri(0, 6)
    x = torch.randn(B, V, generator=g)
    if kind == 0:
        x = x * rf(0.2, 6.0)
    elif kind == 1:
        x = x * rf(5.0, 30.0)
    elif kind == 2:
        x = (x * rf(1.0, 4.0)).round() / ri(1, 4)
    elif kind == 3:
        x = x * rf(0.5, 4.0)
        x[torch.rand(B, V, generator=g) < rf(0.3, 0.995)] = float("-inf")
    elif kind == 4:
        x = x * rf(0.5, 4.0)
        for val in (float("inf"), float("nan"), float("-inf")):
            x[torch.rand(B, V, generator=g) < 0.001] = val
    elif kind == 5:
        x = torch.full((B, V), rf(-5.0, 5.0))
        x[:, torch.randint(0, V, (ri(1, 40),), generator=g)] += rf(0.0, 8.0)
    else:
        x = x * 1e-3 + rf(-100.0, 100.0)
    logits = x.bfloat16().cuda()
    top_k = [1, 2, 5, 50, 64, 100, 128, ri(1, 128)][ri(0, 7)]
    top_p = [1.0, 0.9, 0.95, 0.5, rf(0.05, 1.0)][ri(0, 4)]
    temp = [1.0, 0.6, 0.7, rf(0.1, 2.0)][ri(0, 3)]
    a, b, c = Sampler(B, V, max_new=24), Sampler(B, V, max_new=24, tap=False), Sampler(B, V, max_new=24, tap=False, split=True)
    seed = ri(0, 2 ** 40)
    for s in (a, b, c):
        s.seed.fill_(seed)
    for _ in range(24):
        for s in (a, b, c):
            s(logits, temperature=temp, top_p=top_p, top_k=top_k)
    ok = torch.equal(a.out, b.out) and torch.equal(a.out, c.out) and int(c.ws[:B * 16].view(torch.int32).abs().sum()) == 0
    if not ok:
        bad += 1
        print("MISMATCH case", case, dict(V=V, B=B, kind=kind, top_k=top_k, top_p=top_p, temp=temp, seed=seed), (a.out != b.out).sum().item(), (a.out != c.out).sum().item())
print(f"{cases} cases, {bad} mismatches")
