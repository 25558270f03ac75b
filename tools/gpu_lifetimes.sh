#!/bin/bash
# wave lifetimes of k_nn_tiled in the bench's steady state (last launch of the 40-iteration run), under the knobs given as arguments
for kv in "$@"; do
  echo "== $kv"
  env $kv MOLA_ICP_DEBUG_STATS=2 timeout 300 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --e2e 0 --batch-pairs 0 --shipped-iters 0 2>&1 | grep "mola_icp debug" | tail -n 17
done
