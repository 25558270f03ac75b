#!/bin/bash
# kernel timeline of one steady-state scan of the odometry stream (tools/prof_odometry_stream.py): bash tools/rocprof_odometry.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_odometry; rm -rf $OUT/trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/prof_odometry_stream.py 24 ${DECIMATE:-1} > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - <<PY > $OUT/timeline.txt
import csv
rows = [r for r in csv.DictReader(open("$OUT/kernel_trace.csv"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void mola_icp_amd::", "").replace("mola_icp_amd::", "")[:56]
# the last scan: from the last k_bbox_partial on
start = max(i for i, r in enumerate(rows) if "k_bbox_rows" in r["Kernel_Name"] or "k_bbox_partial" in r["Kernel_Name"])
scan = rows[start:]
t0 = int(scan[0]["Start_Timestamp"]); prev = None; busy = 0.0
print("the last scan of the drive (upload excluded: the first kernel is the new cloud's bounding box)")
for r in scan:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    busy += (e - s) / 1e3
    print("%8.1f us  gap %5.1f  run %6.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, name(r)))
    prev = e
print("span %.1f us, kernels %.1f us, %d launches" % ((prev - t0) / 1e3, busy, len(scan)))
PY
tail -4 $OUT/timeline.txt
