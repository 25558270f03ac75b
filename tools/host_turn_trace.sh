#!/bin/bash
# Where the host turn of an iteration goes (headline config): rocprofv3 --kernel-trace --hip-runtime-trace (no counters) of three
# 40-iteration aligns on 1M x 1M; per iteration: row reduction's end -> hipLaunchKernel entered (flag seen, solve, bookkeeping),
# the call itself, call returned -> matcher starts.     bash tools/host_turn_trace.sh [n] [m]
set -u
N=${1:-1000000}; M=${2:-1000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/host_turn; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/host_turn_run.py <<PY
import importlib, os, sys
import numpy as np
sys.path.insert(0, "$ROOT")
pkg = importlib.import_module("mola-fe-lidar_amd"); synth = importlib.import_module("mola-fe-lidar_amd.synth")
g, l, _ = synth.make_pair($N, $M, seed=42)
icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
p = pkg.Parameters(); p.max_iterations, p.matcher_threshold, p.fixed_iterations, p.skip_quality = 40, 1.0, 1, 1
for k in range(4): icp.align_resident(np.eye(4), p)
PY
rm -rf $OUT/trace
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/trace -- python3 /tmp/host_turn_run.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
K=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); A=$(find $OUT/trace -name "*hip_api_trace.csv" | head -1)
python3 - "$K" "$A" <<'PY' | tee $OUT/host_turn.txt
import csv, sys, statistics
ks = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
api = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
launch = [a for a in api if "LaunchKernel" in a["Function"]]
by_corr = {a["Correlation_Id"]: a for a in launch}
rows = []
prev_end, prev_name = None, None
for k in ks:
    nm = k["Kernel_Name"]
    s, e = int(k["Start_Timestamp"]), int(k["End_Timestamp"])
    a = by_corr.get(k["Correlation_Id"])
    if a and prev_end and "k_nn_tiled" in nm and "k_reduce_items" in (prev_name or ""):
        a0, a1 = int(a["Start_Timestamp"]), int(a["End_Timestamp"])
        rows.append(((a0 - prev_end) / 1e3, (a1 - a0) / 1e3, (s - a1) / 1e3, (s - prev_end) / 1e3))
    prev_end, prev_name = e, nm
rows = rows[len(rows) // 4:]   # (the first align warms up)
md = lambda i: statistics.median(r[i] for r in rows)
print("%d host turns (k_reduce_items end -> next k_nn_tiled start), medians in us:" % len(rows))
print("  reduction ended -> hipLaunchKernel entered (flag seen, Horn, stall test, bookkeeping)  %.2f" % md(0))
print("  inside hipLaunchKernel                                                               %.2f" % md(1))
print("  hipLaunchKernel returned -> kernel started (negative: started before the call returned) %.2f" % md(2))
print("  whole turn                                                                            %.2f   (p90 %.2f)" % (md(3), sorted(r[3] for r in rows)[len(rows) * 9 // 10]))
other = {}
for a in api:
    other.setdefault(a["Function"], []).append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
print("HIP calls of the run: " + ", ".join("%s x%d (median %.1f us)" % (f, len(v), statistics.median(v)) for f, v in sorted(other.items(), key=lambda kv: -len(kv[1]))[:8]))
PY
rm -rf $OUT/trace
