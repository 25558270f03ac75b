#!/bin/bash
# kernel-trace of the shipped pipeline (20 iterations at C3) + instruction counters of the kNN kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${1:-p2pl}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/prof_p2pl.py --iters 20 > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
cut -c1-150 $OUT/kernel_stats.csv | head -12
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/kernel_trace.csv")) if "k_knn_planes" in r["Kernel_Name"] or "k_accumulate_planes" in r["Kernel_Name"]]
for r in rows[-50:]:
    print(r["Kernel_Name"].split("(")[0][-28:], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "us")
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/q -- python3 $ROOT/tools/prof_p2pl.py --iters 20 > $OUT/q.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/r -- python3 $ROOT/tools/prof_p2pl.py --iters 20 > $OUT/r.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/q $OUT/r --kernel k_knn_planes
