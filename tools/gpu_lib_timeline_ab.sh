#!/bin/bash
# same-lease A/B of library builds by the matcher launches of the odometry stream under rocprofv3 (kernel trace):
#   tools/gpu_lib_timeline_ab.sh <tag> a.so b.so ...      (variants in mola-fe-lidar_amd/lib/variants/)
# prints, per variant, the mean duration of the 1st / 2nd / 3rd / 4th k_knn_coop launch behind a scan's bootstrap.
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/mola-fe-lidar_amd/lib; OUT=$ROOT/gpurun_out/$tag; mkdir -p $OUT; cp $L/libmola_icp_amd.so /tmp/orig_lib.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  n=$(basename $v .so)
  cp $L/variants/$n.so $L/libmola_icp_amd.so
  rm -rf $OUT/trace_$n
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$n -- python3 $ROOT/tools/prof_odometry_stream.py 24 > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; cp /tmp/orig_lib.so $L/libmola_icp_amd.so; exit 1; }
  f=$(find $OUT/trace_$n -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$n" <<'PY' | tee $OUT/$n.txt
import csv, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ordn, per, other = None, {}, {}
for r in rows:
    nm = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_bootstrap_seeds" in nm: ordn = 0
    elif "k_knn_coop" in nm and ordn is not None:
        ordn += 1; per.setdefault(ordn, []).append(d)
    else:
        other.setdefault(nm.split("(")[0].replace("void mola_icp_amd::", "")[:40], []).append(d)
print("== %s: k_knn_coop launches behind a bootstrap, mean us (count): %s" % (sys.argv[2], "  ".join("#%d %.1f (%d)" % (k, statistics.mean(v), len(v)) for k, v in sorted(per.items()) if k <= 6)))
PY
  grep "^pass 3" $OUT/$n.log | cut -c1-200
  rm -rf $OUT/trace_$n
done
cp /tmp/orig_lib.so $L/libmola_icp_amd.so
