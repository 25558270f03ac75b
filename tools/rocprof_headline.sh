#!/bin/bash
# rocprofv3 passes over the headline matcher (k_nn_tiled at 1M x 1M, tools/prof_nn.py): bash tools/rocprof_headline.sh <tag>
# Kernel-trace/stats and each PMC set are SEPARATE runs (never combined with other trace domains); FETCH_SIZE and
# WRITE_SIZE in separate passes (TCC counter budget, MI355X_MICROARCH.md).  Afterwards, in the repository:
#   python tools/pmc_record.py gpurun_out/prof_<tag> profiles/r02     -> profiles/r02/counters.json (stamped)
set -u
TAG=${1:-headline}; N=${2:-1000000}; M=${3:-1000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/tools/prof_nn.py --kernel tiled --reps 4 --n $N --m $M > $OUT/$n.log 2>&1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/prof_nn.py --kernel tiled --reps 20 --n $N --m $M > $OUT/trace.log 2>&1
run fetch FETCH_SIZE
run write WRITE_SIZE
run h GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
run g SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run a TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
python3 $ROOT/tools/pmc_summary.py $OUT --kernel k_nn --json $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
cut -c1-200 $OUT/kernel_stats.csv | head -8
echo "$N $M" > $OUT/workload.txt
