#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_config0; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/prof_config0.py > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last resident 8-iteration run: the final ~60 kernels
sel=[r for r in rows if any(k in r["Kernel_Name"] for k in ("k_knn","k_accumulate_planes","k_reduce_rows","k_publish","k_order"))]
t0=None
for r in sel[-40:]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if t0 is None: t0=s
    print("%8.1f us  +%7.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, r["Kernel_Name"].split("(")[0].replace("void mola_icp_amd::","").replace("mola_icp_amd::","")))
PY
