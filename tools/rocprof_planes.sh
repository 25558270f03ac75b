#!/bin/bash
# round 3: rocprofv3 passes over the SHIPPED pipeline's matcher (Matcher_Point2Plane, icp-settings-regular.yaml:33-39):
#   bash tools/rocprof_planes.sh <tag> [n]        n = 1000000: k_knn_planes (unseeded / seeded-dense / counting flavours);
#                                                 n = 120000 (any n <= 131k): k_knn_coop
# Kernel-trace/stats and each PMC set are SEPARATE runs; FETCH_SIZE and WRITE_SIZE in separate passes.  Afterwards, in the repo:
#   python tools/pmc_record_planes.py gpurun_out/prof_<tag> profiles/r03
set -u
TAG=${1:-planes}; N=${2:-1000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IT=20
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/tools/prof_p2pl.py --n $N --iters $IT > $OUT/$n.log 2>&1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/prof_p2pl.py --n $N --iters $IT > $OUT/trace.log 2>&1
run fetch FETCH_SIZE
run write WRITE_SIZE
run h GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
run g SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
python3 $ROOT/tools/pmc_summary.py $OUT --kernel k_ --json $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
cut -c1-220 $OUT/kernel_stats.csv | head -12
echo "$N $N" > $OUT/workload.txt
tail -2 $OUT/trace.log
# timeline of the second (20-iteration) align
python3 - <<PY > $OUT/timeline.txt
import csv
rows=[r for r in csv.DictReader(open("$OUT/kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sel=[r for r in rows if any(k in r["Kernel_Name"] for k in ("k_knn","k_accumulate_planes","k_reduce_rows","k_order","k_publish"))][-100:]
prev=None
for r in sel:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("gap %6.1f us  run %7.1f us  %s" % ((s-prev)/1e3 if prev else 0, (e-s)/1e3, r["Kernel_Name"].split("(")[0].replace("void mola_icp_amd::","").replace("mola_icp_amd::","")[:52]))
    prev=e
PY
tail -45 $OUT/timeline.txt
