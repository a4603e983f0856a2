#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc CSV dumps (FETCH_SIZE, WRITE_SIZE; separate passes) into an entry of
profiles/r1_pmc_traffic.json:  HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE doubled as
MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes for 16-B-per-lane streaming reads on gfx950.
usage: pmc_traffic.py <key> <kernel substring> <fetch.csv> <write.csv> <algorithmic bytes> <json> [M N K]"""
import csv
import json
import sys

key, sub, fcsv, wcsv, alg, jpath = sys.argv[1:7]
shape = [int(a) for a in sys.argv[7:10]]


def avg(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    vals = vals[len(vals) // 2:]          # the timed half of the dispatches (the first pass warms up / faults pages in)
    return sum(vals) / len(vals), len(vals)


f, nf = avg(fcsv, "FETCH_SIZE")
w, _ = avg(wcsv, "WRITE_SIZE")
try:
    d = json.load(open(jpath))
except FileNotFoundError:
    d = {}
d[key] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "kernel": sub, "dispatches_averaged": nf,
          "hbm_bytes_per_launch": (2 * f + w) * 1024, "algorithmic_bytes": int(alg),
          "shape": dict(zip("MNK", shape), dtype="bf16"),
          "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no trace domains) over tools/gemm_one.py; timed half of the dispatches (" + __import__("os").environ.get("LL_PMC_ROUND", "r5") + ")"}
json.dump(d, open(jpath, "w"), indent=1)
print(key, d[key]["hbm_bytes_per_launch"], "vs algorithmic", alg)
