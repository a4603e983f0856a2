#!/usr/bin/env python3
"""Device timeline of ONE iteration out of a rocprofv3 (rocpd sqlite) kernel trace: every launch between two occurrences of an
anchor kernel, with its start offset, duration and the idle gap before it (us).

usage: tools/rocpd_timeline.py results.db <anchor substring> [which occurrence, default: the middle one] [out.csv]
"""
import sqlite3
import sys

from rocpd_stats import short


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    seq = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    anchor = sys.argv[2]
    idx = [i for i, r in enumerate(seq) if anchor in r[0]]
    if len(idx) < 2:
        sys.exit(f"anchor {anchor!r} occurs {len(idx)} times")
    k = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else len(idx) // 2
    a, b = idx[k], idx[k + 1] if k + 1 < len(idx) else len(seq)
    t0 = seq[a][1]
    lines = ["i,kernel,start_us,dur_us,gap_before_us"]
    prev_end = None
    busy = 0
    for i in range(a, b):
        n, s, e = seq[i]
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        lines.append(f"{i - a},\"{short(n)}\",{(s - t0) / 1e3:.2f},{(e - s) / 1e3:.2f},{gap:.2f}")
        busy += e - s
        prev_end = e
    span = (seq[b - 1][2] - t0) / 1e3
    lines.append(f"# {b - a} launches, span {span:.1f} us, kernel time {busy / 1e3:.1f} us, idle {span - busy / 1e3:.1f} us; "
                 f"next anchor starts {((seq[b][1] - t0) / 1e3) if b < len(seq) else float('nan'):.1f} us after this one")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(out + "\n")


if __name__ == "__main__":
    main()
