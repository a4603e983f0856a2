#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max (us).

usage: tools/rocpd_stats.py results.db [out.csv]
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)", "anon")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("ll::", "")
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, (end - start) from kernels").fetchall()
    if len(sys.argv) > 3:      # third argument: also write the average idle gap AFTER each kernel type (device timeline order)
        seq = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
        gaps = {}
        for (n0, s0, e0), (n1, s1, e1) in zip(seq, seq[1:]):
            g = s1 - e0
            if 0 <= g < 50000:     # ignore host-side pauses
                a = gaps.setdefault(short(n0), [0, 0])
                a[0] += 1
                a[1] += g
        with open(sys.argv[3], "w") as f:
            f.write("kernel,n,avg_gap_after_us\n")
            for n, a in sorted(gaps.items(), key=lambda kv: -kv[1][1]):
                f.write(f"\"{n}\",{a[0]},{a[1]/a[0]/1e3:.2f}\n")
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0, 1e30, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,pct"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{n}\",{a[0]},{a[1]/1e3:.1f},{a[1]/a[0]/1e3:.2f},{a[2]/1e3:.2f},{a[3]/1e3:.2f},{100*a[1]/tot:.1f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
