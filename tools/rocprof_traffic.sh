#!/bin/bash
# HBM traffic of the NN kernels (separate --pmc passes, as the guide prescribes) + kernel-trace stats of bench.py.
#   bash tools/rocprof_traffic.sh <tag>     -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in tiled mfma; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$K -- python3 $ROOT/tools/prof_nn.py --kernel $K --reps 3 > $OUT/fetch_$K.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$K -- python3 $ROOT/tools/prof_nn.py --kernel $K --reps 3 > $OUT/write_$K.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $ROOT/bench.py --cpu-baseline-iters 0 > $OUT/bench_trace.log 2>&1
