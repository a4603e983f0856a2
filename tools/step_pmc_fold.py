#!/usr/bin/env python3
"""Fold the three whole-trajectory PMC passes of tools/profile_step_pmc.sh (MfmaUtil, FETCH_SIZE, WRITE_SIZE; one counter
per pass) into per-kernel and per-reverse-step figures: HBM bytes per step = (2*FETCH_SIZE + WRITE_SIZE) * 1024 summed over
the kernels of one timed trajectory / T (FETCH doubled per MI355X_MICROARCH.md, HBM section), MfmaUtil per kernel type.
usage: step_pmc_fold.py <tag> mfma.csv fetch.csv write.csv out.json [bench args]"""
import csv
import json
import re
import sys

tag, mcsv, fcsv, wcsv, outp = sys.argv[1:6]
args = sys.argv[6:]
T = 50
B = int(args[args.index("--batch") + 1]) if "--batch" in args else 8


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("ll::", "")[:90]


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    return [(short(r["Kernel_Name"]), float(r["Counter_Value"])) for r in rows]


def trajectory(rows):
    """dispatches of the LAST trajectory (warm-up + timed run the same kernels): second half of the denoiser's dispatches"""
    idx = [i for i, (k, _) in enumerate(rows) if k.startswith("advance_step_kernel")]
    if len(idx) >= 2 * T:
        start = idx[len(idx) - T - 1] + 1     # right after the last step of the previous trajectory
        return rows[start:idx[-1] + 1]
    return rows[len(rows) // 2:]


res = {"tag": tag, "batch": B, "T": T, "kernels": {}}
for path, ctr in ((mcsv, "MfmaUtil"), (fcsv, "FETCH_SIZE"), (wcsv, "WRITE_SIZE")):
    try:
        rows = trajectory(load(path, ctr))
    except FileNotFoundError:
        continue
    for k, v in rows:
        d = res["kernels"].setdefault(k, {})
        d.setdefault(ctr, [0, 0.0])
        d[ctr][0] += 1
        d[ctr][1] += v
tot_f = tot_w = 0.0
for k, d in res["kernels"].items():
    for ctr in list(d):
        n, s = d[ctr]
        d[ctr] = {"dispatches": n, "avg": s / n, "sum": s}
    tot_f += d.get("FETCH_SIZE", {}).get("sum", 0.0)
    tot_w += d.get("WRITE_SIZE", {}).get("sum", 0.0)
res["hbm_bytes_per_step"] = (2 * tot_f + tot_w) * 1024 / T
res["fetch_kb_per_step"] = tot_f / T
res["write_kb_per_step"] = tot_w / T
res["method"] = ("rocprofv3 --pmc <one counter> per pass over bench.py --workload graphdit --no-graph; kernels of the last trajectory; "
                 "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH doubled per MI355X_MICROARCH.md for 16-B/lane streaming reads)")
json.dump(res, open(outp, "w"), indent=1)
print("HBM bytes per reverse step: %.1f MB (fetch %.1f MB x2, write %.1f MB)" % (res["hbm_bytes_per_step"] / 1e6, tot_f * 1024 / T / 1e6, tot_w * 1024 / T / 1e6))
for k, d in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {}).get("sum", 0.0))[:8]:
    print("%-70s MfmaUtil %5.1f  fetch/launch %8.1f KB  x%d" % (k[:70], d.get("MfmaUtil", {}).get("avg", float("nan")),
                                                                d.get("FETCH_SIZE", {}).get("avg", float("nan")), d.get("FETCH_SIZE", {}).get("dispatches", 0)))
