#!/bin/bash
# instruction counts of k_nn_tiled at C3 with 64-query items (MOLA_ICP_QPL=1) vs the default
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_qpl1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for q in 1 2; do
  export MOLA_ICP_QPL=$q
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/q$q -- python3 $ROOT/tools/prof_nn.py --kernel tiled --reps 4 > $OUT/q$q.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/r$q -- python3 $ROOT/tools/prof_nn.py --kernel tiled --reps 4 > $OUT/r$q.log 2>&1
  echo "QPL=$q"; python3 $ROOT/tools/pmc_summary.py $OUT/q$q $OUT/r$q --kernel "k_nn_tiled<false"
done
