#!/bin/bash
# average duration of the named kernels over one run of the odometry stream under rocprofv3 (for library A/B runs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(mktemp -d /tmp/kt.XXXXXX)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/prof_odometry_stream.py 24 > $OUT/log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void mola_icp_amd::", "").replace("mola_icp_amd::", "")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
import statistics
for n in sorted(d):
    if any(k in n for k in ("$1".split(","))):
        print("%-40s n %5d  median %7.2f us  mean %7.2f" % (n[:40], len(d[n]), statistics.median(d[n]), sum(d[n]) / len(d[n])))
PY
