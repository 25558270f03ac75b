#!/usr/bin/env python3
"""Turns the output of tools/rocprof_planes.sh into stamped records of profiles/<round>/counters.json -- one per flavour of the
plane matcher (kernel symbol), plus a `k_knn_planes` record (launch-weighted average over the flavours of the 20-iteration
run) that bench.py attaches to `shipped_point2plane_gn.roofline`.  Usage: pmc_record_planes.py <gpurun_out/prof_tag> <profiles/rNN>"""
import csv
import datetime
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha1  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
summ = json.load(open(os.path.join(src, "pmc_summary.json")))
n, m = (int(x) for x in open(os.path.join(src, "workload.txt")).read().split())
# average duration per kernel symbol from the kernel trace of the same command
dur = {}
with open(os.path.join(src, "kernel_trace.csv")) as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"].split("(")[0]
        dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
stamp = {"commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
         "date": datetime.date.today().isoformat(), "kernel_sources_sha1": kernel_sources_sha1()}
K = 6
bytes_alg = 12.0 * n + 12.0 * m + (112.0 + 4.0 * (K + 1) + 4.0) * n
recs_new = []
tot = {"launches": 0, "us": 0.0, "hbm": 0.0, "valu": 0.0}
for sym, cs in summ.items():
    if "k_knn" not in sym:
        continue
    c = {name: v["mean"] for name, v in cs.items()}
    nl = int(cs.get("SQ_INSTS_VALU", next(iter(cs.values())))["n"])
    d = {}
    full = [k for k in dur if k.startswith(sym.rstrip("."))] or [k for k in dur if sym[:40] in k]
    us = sum(dur[full[0]]) / len(dur[full[0]]) if full else None
    if us:
        d["avg_us"] = us
        d["launches_in_trace"] = len(dur[full[0]])
        # (no HBM fraction per flavour: a launch average weighs the verification / counting launches of microseconds like full
        # searches -- bench.py prices the matcher per ITERATION, shipped_point2plane_gn.roofline)
    if "GRBM_GUI_ACTIVE" in c:
        d["kernel_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * d["kernel_cycles_per_xcd"])
    if "SQ_WAVE_CYCLES" in c:
        d["wave_cycles_waiting_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        d["wave_cycles_issuing_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in c:
        d["valu_insts_per_64_query_item"] = c["SQ_INSTS_VALU"] / ((n + 63) // 64)
    rec = {"kernel": sym, "n_local": n, "n_map": m, "counters_mean_per_launch": c, "launches_per_pass": nl, "derived": d, **stamp,
           "note": "mean over the launches of this flavour in each --pmc pass of tools/prof_p2pl.py --iters 20 (a 1-iteration align, then a "
                   "20-iteration one); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), FETCH_SIZE / WRITE_SIZE in KiB"}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rec["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / bytes_alg
    recs_new.append(rec)
    if us:
        tot["launches"] += len(dur[full[0]]); tot["us"] += sum(dur[full[0]])
        tot["hbm"] += rec.get("hbm_bytes_per_launch", 0.0) * len(dur[full[0]])
if tot["launches"]:
    avg_us = tot["us"] / tot["launches"]
    recs_new.append({"kernel": "k_knn_planes" if n > 131072 else "k_knn_coop", "n_local": n, "n_map": m, **stamp,
                     "hbm_bytes_per_launch": tot["hbm"] / tot["launches"],
                     "derived": {"avg_us_over_all_flavours": avg_us, "launches": tot["launches"], "algorithmic_bytes_per_launch": bytes_alg,
                                 },
                     "note": "launch-weighted average over the matcher flavours of the run (see the per-symbol records)"})
os.makedirs(dst, exist_ok=True)
path = os.path.join(dst, "counters.json")
recs = json.load(open(path)) if os.path.exists(path) else []
keys = {(r["kernel"], r["n_local"], r["n_map"]) for r in recs_new}
recs = [r for r in recs if (r["kernel"], r["n_local"], r["n_map"]) not in keys] + recs_new
json.dump(recs, open(path, "w"), indent=1)
tag = f"planes_{n}x{m}"
for f, name in (("kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("pmc_summary.txt", f"{tag}_pmc_summary.txt"), ("timeline.txt", f"{tag}_kernel_trace.txt")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, name))
for r in recs_new:
    print(r["kernel"][:60], json.dumps(r["derived"]), r.get("hbm_bytes_per_launch"))
