#!/bin/bash
# kernel timeline + per-kernel totals of the loop-closure Monte-Carlo batch: bash tools/rocprof_mc.sh [p2pl|p2p] [n]
set -u
PIPE=${1:-p2pl}; N=${2:-100000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_mc_$PIPE; mkdir -p $OUT
python3 $ROOT/tools/prof_mc.py --pipeline $PIPE --n $N > $OUT/plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/prof_mc.py --pipeline $PIPE --n $N --reps 1 > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - <<PY > $OUT/timeline.txt
import csv, collections
rows=[r for r in csv.DictReader(open("$OUT/kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
name=lambda r: r["Kernel_Name"].split("(")[0].replace("void mola_icp_amd::","").replace("mola_icp_amd::","")[:48]
# the last multi_init call: from the last k_bbox / sort kernels on
last_prep=max(i for i,r in enumerate(rows) if "k_tile_boxes" in r["Kernel_Name"] or "k_super_boxes" in r["Kernel_Name"])
call=rows[last_prep+1:]
tot=collections.defaultdict(lambda:[0,0.0])
prev=None; gaps=0.0
for r in call:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    tot[name(r)][0]+=1; tot[name(r)][1]+=(e-s)/1e3
    if prev: gaps+=max(0,(s-prev)/1e3)
    prev=e
span=(int(call[-1]["End_Timestamp"])-int(call[0]["Start_Timestamp"]))/1e3
print("last call after its preparation: %d kernels, span %.1f us, gaps %.1f us" % (len(call), span, gaps))
for k,(n,t) in sorted(tot.items(), key=lambda kv:-kv[1][1]): print("  %-48s x%-4d %9.1f us  (avg %.1f)" % (k,n,t,t/n))
prev=None
for r in call[:64]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("gap %6.1f us  run %7.1f us  grid %-8s %s" % ((s-prev)/1e3 if prev else 0, (e-s)/1e3, r.get("Grid_Size_X","?")+"x"+r.get("Grid_Size_Y","?"), name(r)))
    prev=e
PY
cat $OUT/plain.log | tail -4; head -70 $OUT/timeline.txt
