#!/bin/bash
# counters of the point-to-plane matcher kernel: bash tools/rocprof_p2pl.sh <tag>
set -u
TAG=${1:-p2pl}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/a -- python3 $ROOT/tools/prof_p2pl.py --iters 6 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k -- python3 $ROOT/tools/prof_p2pl.py --iters 6 > $OUT/k.log 2>&1
