#!/bin/bash
# quick PMC look at the NN kernel: bash tools/rocprof_quick.sh <tag>
set -u
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $ROOT/tools/prof_nn.py --kernel ${KERN:-mfma} --reps 3 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc2 -- python3 $ROOT/tools/prof_nn.py --kernel ${KERN:-mfma} --reps 3 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_WAVES_RESTORED SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc3 -- python3 $ROOT/tools/prof_nn.py --kernel ${KERN:-mfma} --reps 3 > $OUT/pmc3.log 2>&1
