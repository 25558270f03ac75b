#!/bin/bash
# plane-matcher kernel time per scan of the odometry stream under rocprofv3 --kernel-trace, per library variant (device durations:
# what a same-lease A/B of wall times cannot resolve):  bash tools/gpu_kq4_trace.sh <out tag> <variant> [...]   ("product" = as built)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = product ]; then unset MOLA_ICP_LIB_PATH; else export MOLA_ICP_LIB_PATH=$ROOT/mola-fe-lidar_amd/lib/variants/$v.so; fi
  rm -rf $O/trace_$v
  MOLA_ICP_KNN_Q4=${KNN_Q4:-1} rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -- python3 $ROOT/tools/prof_odometry_stream.py 24 ${DECIMATE:-1} > $O/trace_$v.log 2>&1
  f=$(find $O/trace_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# scans = stretches that start at a bounding-box kernel; the last pass = the last 23 of them (the first scan of a pass has no ICP)
starts = [i for i, r in enumerate(rows) if "k_bbox_rows" in r["Kernel_Name"] or "k_bbox_partial" in r["Kernel_Name"]]
starts = starts[-23:] + [len(rows)]
per = {}
spans = []
for a, b in zip(starts[:-1], starts[1:]):
    scan = rows[a:b]
    k = [r for r in scan if "k_knn" in r["Kernel_Name"]]
    for j, r in enumerate(k):
        per.setdefault(j, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    spans.append((int(scan[-1]["End_Timestamp"]) - int(scan[0]["Start_Timestamp"])) / 1e3)
med = lambda v: sorted(v)[len(v) // 2]
tot = sum(med(v) for j, v in per.items() if len(v) > 11)
print("%-14s plane-matcher launches per scan (median us over %d scans): %s | sum %.1f | scan span median %.1f us" % (
    sys.argv[2], len(spans), " ".join("%.1f" % med(v) for j, v in sorted(per.items()) if len(v) > 11), tot, med(spans)))
PY
done | tee $O/trace_summary.txt
