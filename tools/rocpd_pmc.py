#!/usr/bin/env python3
"""Per-kernel average of PMC counters from a rocprofv3 (rocpd sqlite) run.  usage: rocpd_pmc.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("ll::", "")
    return name[:110]


db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print("columns:", cols, file=sys.stderr)
name_col = "kernel_name" if "kernel_name" in cols else "name"
cname = "counter_name" if "counter_name" in cols else "pmc_name"
val = "value" if "value" in cols else "counter_value"
rows = cur.execute(f"select {name_col}, {cname}, {val} from counters_collection").fetchall()
agg = {}
for k, c, v in rows:
    a = agg.setdefault((short(k), c), [0, 0.0])
    a[0] += 1
    a[1] += float(v)
lines = ["kernel,counter,dispatches,sum,avg_per_dispatch"]
for (k, c), (n, sv) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"\"{k}\",{c},{n},{sv:.1f},{sv/n:.2f}")
out = "\n".join(lines)
print("\n".join(lines[:14]))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
