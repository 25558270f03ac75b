#!/bin/bash
# kernel timeline (gap to the previous kernel, duration) of a resident align: bash tools/rocprof_timeline_any.sh <n> <p2p|p2pl> [iters]
set -u
N=${1:-100000}; PIPE=${2:-p2p}; IT=${3:-12}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_timeline_any; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tl_any.py <<PY
import importlib, os, sys, numpy as np
sys.path.insert(0, "$ROOT")
pkg = importlib.import_module("mola-fe-lidar_amd"); synth = importlib.import_module("mola-fe-lidar_amd.synth")
g, l, _ = synth.make_pair($N, $N, seed=42)
icp = pkg.ICP(device=0); icp.set_map(g); icp.set_local(l)
if "$PIPE" == "p2pl":
    p = pkg.Parameters.load_from_file(os.path.join("$ROOT", "params", "icp-settings-regular.yaml"))
else:
    p = pkg.Parameters(); p.matcher_threshold = 1.0
p.fixed_iterations, p.skip_quality, p.max_iterations = 1, 1, $IT
icp.align_resident(np.eye(4), p); icp.align_resident(np.eye(4), p)
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 /tmp/tl_any.py > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
prev=None
for r in rows[-30:]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("gap %6.1f us  run %7.1f us  %s" % ((s-prev)/1e3 if prev else 0, (e-s)/1e3, r["Kernel_Name"].split("(")[0].replace("void mola_icp_amd::","").replace("mola_icp_amd::","")[:44]))
    prev=e
PY
