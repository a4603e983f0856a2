#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of the e2e bench: the period between consecutive decode_prologue_kernel launches (= one decode token on the
device), the kernel time of the DECODE kernels inside it, and the idle time, as distributions.  usage: tools/token_periods.py results.db"""
import sqlite3
import statistics as st
import sys

DECODE = ("decode_prologue", "gemv_fused", "gemv_stage", "decode_attn_rope", "gemv_bf16_kernel", "sample_token", "rmsnorm_bf16", "index_select", "embedding", "indexSelect")
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
seq = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(seq) if "decode_prologue" in r[0]]
periods, busy, other, n_k = [], [], [], []
for a, b in zip(idx, idx[1:]):
    p = (seq[b][1] - seq[a][1]) / 1e3
    if p > 4000:        # not consecutive tokens of one prompt
        continue
    d = sum(e - s for n, s, e in seq[a:b] if any(k in n for k in DECODE)) / 1e3
    o = sum(e - s for n, s, e in seq[a:b] if not any(k in n for k in DECODE)) / 1e3
    periods.append(p); busy.append(d); other.append(o); n_k.append(b - a)
q = lambda v: (min(v), st.median(v), st.mean(v), max(v))      # noqa: E731
print(f"{len(periods)} token periods")
print("period us   min/median/mean/max: %.1f %.1f %.1f %.1f" % q(periods))
print("decode kernels us              : %.1f %.1f %.1f %.1f" % q(busy))
print("other kernels inside us        : %.1f %.1f %.1f %.1f" % q(other))
print("launches per period            : %d %d %.1f %d" % q(n_k))
quiet = [(p, d) for p, d, o in zip(periods, busy, other) if o == 0]
if quiet:
    print(f"{len(quiet)} periods with no other kernel: period median {st.median([p for p, _ in quiet]):.1f} us, decode kernels median {st.median([d for _, d in quiet]):.1f} us")

# per molecule: the decode tokens (periods between prologues), what the slow periods cost beyond the median one, and the stretch between the last
# prologue of one prompt and the first of the next (last token + query forward + collection + next prefill)
med = st.median(periods)
starts = [seq[i][1] for i in idx]
mol, cur_p = [], []
for a, b in zip(starts, starts[1:]):
    p = (b - a) / 1e3
    if p > 4000:
        mol.append((cur_p, p))
        cur_p = []
    else:
        cur_p.append(p)
print("per prompt: tokens, sum of periods ms, excess over median ms (periods > median + 100 us: count), boundary stretch ms")
for ps, gap in mol[1:9]:
    slow = [p for p in ps if p > med + 100]
    print(f"  {len(ps):4d}  {sum(ps) / 1e3:8.2f}  {(sum(ps) - med * len(ps)) / 1e3:6.2f} ({len(slow)}: {' '.join('%.0f' % p for p in slow[:10])})  {gap / 1e3:7.2f}")

# what fills one boundary stretch (the 5th): per kernel name calls / total us, and the idle time of the device in it
from rocpd_stats import short      # noqa: E402
bounds = [(a, b) for a, b in zip(idx, idx[1:]) if (seq[b][1] - seq[a][1]) / 1e3 > 4000]
if len(bounds) > 5:
    a, b = bounds[5]
    t0, t1 = seq[a][1], seq[b][1]
    agg, busy_until, idle, big_gaps = {}, seq[a][2], 0, []
    for n, s, e in seq[a:b]:
        x = agg.setdefault(short(n)[:90], [0, 0])
        x[0] += 1
        x[1] += e - s
        if s > busy_until:
            idle += s - busy_until
            if s - busy_until > 200000:
                big_gaps.append(((busy_until - t0) / 1e3, (s - busy_until) / 1e3, short(n)[:60]))
        busy_until = max(busy_until, e)
    print(f"boundary stretch {(t1 - t0) / 1e6:.2f} ms: {b - a} launches, device idle {idle / 1e6:.2f} ms")
    for at, g, n in big_gaps:
        print(f"   idle {g:8.1f} us at +{at:9.1f} us before {n}")
    for n, x in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"   {x[0]:5d} x {x[1] / x[0] / 1e3:9.2f} us = {x[1] / 1e6:7.3f} ms  {n}")
