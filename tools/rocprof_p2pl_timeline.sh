#!/bin/bash
# timeline of the shipped pipeline at C3 (20 fixed iterations): every launch of the second resident align
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_p2pl_timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/prof_p2pl.py --iters 20 > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sel=[r for r in rows if any(k in r["Kernel_Name"] for k in ("k_knn_planes","k_accumulate_planes","k_reduce_rows","k_order"))][-90:]
prev=None
for r in sel:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("gap %6.1f us  run %7.1f us  %s" % ((s-prev)/1e3 if prev else 0, (e-s)/1e3, r["Kernel_Name"].split("(")[0].replace("void mola_icp_amd::","").replace("mola_icp_amd::","")[:40]))
    prev=e
PY
tail -3 $OUT/trace.log
