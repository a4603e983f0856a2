#!/usr/bin/env python3
"""Fold the whole-trajectory PMC passes of tools/profile_step_pmc.sh (one counter per pass: MfmaUtil, FETCH_SIZE, WRITE_SIZE) and the
kernel-trace statistics of the same command into per-kernel and per-reverse-step figures:

  HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (FETCH doubled per MI355X_MICROARCH.md, HBM section: on gfx950 the
                              counter reports half of a 16-B-per-lane streaming read; the counter sits on the fabric side of the eight
                              per-XCD L2s, so Infinity-Cache hits are included)
  GB/s per kernel           = bytes per launch / average kernel duration of the kernel-trace run
  MFMA busy                 = MfmaUtil (percent of the kernel's cycles with the matrix pipe busy; gfx94x formula, see the guide)
  per step                  = sum over the launches of ONE trajectory / T, and the duration-weighted MfmaUtil

The engine launches every kernel from its own loop (`--no-graph`: counter collection does not see kernels inside a replayed hipGraph),
so a trajectory is the dispatches between the last `init_state_kernel` and the second `set_scalars_kernel` after it (ll_dit_run sets the
scalars once before and once after its T steps) -- the micro-benchmark bench.py runs afterwards for its roofline object is cut off.

usage: step_pmc_fold.py <tag> mfma.csv fetch.csv write.csv out.json <kernel_stats.csv | -> <bench.json | -> [bench args]"""
import csv
import json
import re
import sys

tag, mcsv, fcsv, wcsv, outp, stats_csv, bench_json = sys.argv[1:8]
args = sys.argv[8:]
T = int(args[args.index("--T") + 1]) if "--T" in args else 50
B = int(args[args.index("--batch") + 1]) if "--batch" in args else 8
HBM_PEAK, MFMA_PEAK = 8.0e12, 2.5e15


def short(name):
    name = name.replace("(anonymous namespace)", "anon")
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("ll::", "")[:110]


def load(path, counter):
    allrows = list(csv.DictReader(open(path)))
    rows = [r for r in allrows if r["Counter_Name"] == counter]
    if not rows and counter == "FETCH_SIZE":      # the raw counter behind it: FETCH_SIZE [KB] = TCC_EA0_RDREQ x 64 B / 1024 (MI355X_MICROARCH.md)
        rows = [dict(r, Counter_Value=str(float(r["Counter_Value"]) * 64.0 / 1024.0)) for r in allrows if r["Counter_Name"].startswith("TCC_EA0_RDREQ")]
        if rows:
            res_notes.append("FETCH_SIZE taken as TCC_EA0_RDREQ_sum x 64 B (the derived counter's pass crashed rocprofv3 on this kernel mix)")
    key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
    if key:
        rows.sort(key=lambda r: int(r[key]))
    return [(short(r["Kernel_Name"]), float(r["Counter_Value"])) for r in rows]


def trajectory(rows):
    starts = [i for i, (k, _) in enumerate(rows) if k.startswith("init_state_kernel")]
    if not starts:
        return []
    i0 = starts[-1]
    seen = 0
    for j in range(i0 + 1, len(rows)):
        if rows[j][0].startswith("set_scalars_kernel"):
            seen += 1
            if seen == 2:
                return rows[i0 + 1:j]
    return rows[i0 + 1:]


res_notes = []
durations = {}
if stats_csv != "-":
    try:
        for r in csv.DictReader(open(stats_csv)):
            durations[r["kernel"]] = float(r["avg_us"])
    except FileNotFoundError:
        pass
res = {"tag": tag, "batch": B, "T": T, "kernels": {}}
launches = None
for path, ctr in ((mcsv, "MfmaUtil"), (fcsv, "FETCH_SIZE"), (wcsv, "WRITE_SIZE")):
    try:
        rows = trajectory(load(path, ctr))
    except FileNotFoundError:
        continue
    if rows and launches is None:
        launches = len(rows)
    for k, v in rows:
        d = res["kernels"].setdefault(k, {})
        d.setdefault(ctr, [0, 0.0])
        d[ctr][0] += 1
        d[ctr][1] += v
tot_f = tot_w = 0.0
t_sum = mfma_weighted = 0.0
for k, d in res["kernels"].items():
    for ctr in list(d):
        n, s = d[ctr]
        d[ctr] = {"dispatches": n, "avg": s / n, "sum": s}
    f, w = d.get("FETCH_SIZE", {}), d.get("WRITE_SIZE", {})
    tot_f += f.get("sum", 0.0)
    tot_w += w.get("sum", 0.0)
    if f and w:
        d["hbm_side_bytes_per_launch"] = (2 * f["avg"] + w["avg"]) * 1024
    elif f:      # the WRITE_SIZE pass is missing: read side only
        d["hbm_side_read_bytes_per_launch"] = 2 * f["avg"] * 1024
    n = (f or w or d.get("MfmaUtil", {})).get("dispatches", 0)
    d["launches_per_step"] = n / T
    us = durations.get(k)
    if us is None:      # the statistics file truncates names at the same length: match by prefix
        us = next((v for kk, v in durations.items() if kk.startswith(k[:60]) or k.startswith(kk[:60])), None)
    if us is not None:
        d["avg_us"] = us
        bl = d.get("hbm_side_bytes_per_launch", d.get("hbm_side_read_bytes_per_launch"))
        if bl is not None:
            d["hbm_side_GBps"] = bl / (us * 1e-6) / 1e9
            d["hbm_side_frac_of_peak"] = d["hbm_side_GBps"] * 1e9 / HBM_PEAK
        t_sum += us * n
        mfma_weighted += us * n * d.get("MfmaUtil", {}).get("avg", 0.0)
import os
if tot_w == 0.0 and os.environ.get("WRITE_KB_PER_STEP"):
    # the WRITE_SIZE pass of THIS collection crashed; total of an earlier successful pass over the same command (per-kernel split not kept)
    tot_w = float(os.environ["WRITE_KB_PER_STEP"]) * T
    res_notes.append("WRITE_SIZE per step from an earlier successful pass over the same command (its per-kernel split was not kept; "
                     "five later passes crashed rocprofv3): %.1f KB" % float(os.environ["WRITE_KB_PER_STEP"]))
res["launches_per_step"] = (launches or 0) / T
res["hbm_side_bytes_per_step"] = (2 * tot_f + tot_w) * 1024 / T
res["fetch_kb_per_step"] = tot_f / T
res["write_kb_per_step"] = tot_w / T
if t_sum > 0:
    res["kernel_time_us_per_step"] = t_sum / T
    res["mfma_busy_percent_time_weighted"] = mfma_weighted / t_sum
if bench_json != "-":
    try:
        b = json.loads(open(bench_json).read().strip().splitlines()[-1])
        res["denoise_step_ms"] = b["denoise_step_ms"]
        res["algorithmic_bytes_per_step"] = b["step_roofline"]["hbm_bytes"]
        res["algorithmic_flops_per_step"] = b["step_roofline"]["flops"]
        st = b["denoise_step_ms"] * 1e-3
        res["step_hbm_side_GBps"] = res["hbm_side_bytes_per_step"] / st / 1e9
        res["step_hbm_side_frac_of_peak"] = res["hbm_side_bytes_per_step"] / st / HBM_PEAK
        res["step_algorithmic_hbm_frac_of_peak"] = b["step_roofline"]["hbm_frac"]
        res["step_mfma_frac_of_peak"] = b["step_roofline"]["mfma_frac"]
        res["traffic_over_algorithmic"] = res["hbm_side_bytes_per_step"] / b["step_roofline"]["hbm_bytes"]
    except Exception as e:      # noqa: BLE001
        res["bench_json_error"] = str(e)
res["method"] = ("rocprofv3 --pmc <one counter> per pass (no trace domains) over `python3 bench.py --workload graphdit --no-graph --steps 1 --warmup 1`; "
                 "kernels of the last trajectory (init_state_kernel .. second set_scalars_kernel); durations from the rocprofv3 --kernel-trace --stats "
                 "run of the same command; HBM-side bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH doubled per MI355X_MICROARCH.md)")
if res_notes:
    res["notes"] = res_notes
json.dump(res, open(outp, "w"), indent=1)
print("launches per step %.1f; HBM-side bytes per reverse step: %.1f MB (fetch %.1f MB x2, write %.1f MB); MFMA busy (time-weighted) %.1f %%" %
      (res["launches_per_step"], res["hbm_side_bytes_per_step"] / 1e6, tot_f * 1024 / T / 1e6, tot_w * 1024 / T / 1e6,
       res.get("mfma_busy_percent_time_weighted", float("nan"))))
for k, d in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {}).get("sum", 0.0))[:10]:
    print("%-64s x%5.1f/step  MfmaUtil %5.1f  %8.1f KB/launch  %6.2f us  %7.1f GB/s" % (
        k[:64], d["launches_per_step"], d.get("MfmaUtil", {}).get("avg", float("nan")),
        d.get("hbm_side_bytes_per_launch", d.get("hbm_side_read_bytes_per_launch", float("nan"))) / 1024,
        d.get("avg_us", float("nan")), d.get("hbm_side_GBps", float("nan"))))
