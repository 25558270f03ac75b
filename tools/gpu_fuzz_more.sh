#!/bin/bash
# more of tests/test_gpu_fuzz.py than the suite runs: chunks of random cases under different seeds (a progress line per chunk)
#   bash tools/gpu_fuzz_more.sh [chunks] [cases per chunk] [large cases per chunk]
set -o pipefail
CH=${1:-4}; CASES=${2:-150}; LARGE=${3:-10}
O=gpurun_out/fuzz_more; mkdir -p $O
for s in $(seq 1 $CH); do
  MOLA_ICP_FUZZ_SEED=$((${SEED0:-1000} + s)) MOLA_ICP_FUZZ_CASES=$CASES MOLA_ICP_FUZZ_LARGE_CASES=$LARGE timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/chunk_$s.log 2>&1
  rc=$?; echo "chunk $s (seed $((${SEED0:-1000} + s)), $CASES + $LARGE cases): $(tail -1 $O/chunk_$s.log)"
  [ $rc -ne 0 ] && { tail -30 $O/chunk_$s.log; exit $rc; }
done
exit 0
