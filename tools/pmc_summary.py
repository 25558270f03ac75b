#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output: mean counter value per dispatch, per kernel (name prefix), over every
*counter_collection.csv below the given directories.  Usage: pmc_summary.py <dir> [<dir> ...] [--kernel SUBSTR] [--json OUT]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

args = sys.argv[1:]
kern = None
out_json = None
dirs = []
i = 0
while i < len(args):
    if args[i] == "--kernel":
        kern = args[i + 1]; i += 2
    elif args[i] == "--json":
        out_json = args[i + 1]; i += 2
    else:
        dirs.append(args[i]); i += 1
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = k
        for (disp, cname), v in per_dispatch.items():
            acc[names[disp]][cname].append(v)
res = {}
for k, cs in sorted(acc.items()):
    if kern and kern not in k:
        continue
    short = k if len(k) < 70 else k[:67] + "..."
    res[short] = {c: {"mean": sum(v) / len(v), "n": len(v), "last": v[-1]} for c, v in sorted(cs.items())}
    print(short)
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} mean {sum(v)/len(v):16.1f}  last {v[-1]:16.1f}  (n={len(v)})")
if out_json:
    with open(out_json, "w") as fh:
        json.dump(res, fh, indent=1)
