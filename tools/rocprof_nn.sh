#!/bin/bash
# rocprofv3 passes over the NN matcher alone (tools/prof_nn.py).  Run on the GPU box:
#   bash tools/rocprof_nn.sh <kernel: mfma|valu> <tag>
# Kernel-trace/stats and each PMC set are SEPARATE runs (never combined with other trace domains).
set -u
K=${1:-mfma}; TAG=${2:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${K}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/prof_nn.py --kernel $K --reps 5 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $ROOT/tools/prof_nn.py --kernel $K --reps 2 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 $ROOT/tools/prof_nn.py --kernel $K --reps 2 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc3 -- python3 $ROOT/tools/prof_nn.py --kernel $K --reps 2 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc4 -- python3 $ROOT/tools/prof_nn.py --kernel $K --reps 2 > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
