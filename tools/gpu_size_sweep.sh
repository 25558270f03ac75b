#!/bin/bash
# k_nn_tiled: item size x blocks per CU over cloud sizes (headline bench at --n-local / --n-map)
line() { python -c "
import json,sys; d=json.load(sys.stdin)
print('$1', 'value %.0f it/s  kernel %.1f us  pairs/query %.0f' % (d['value'], d['roofline']['kernel_ms']*1e3, d['roofline']['flop_view']['pairs_evaluated_per_query']))"; }
for nm in "400000 400000" "600000 600000" "1000000 1000000" "2000000 2000000" "1000000 10000000"; do
  set -- $nm
  for kv in "MOLA_ICP_QPL=2 MOLA_ICP_BLOCKS_PER_CU=3" "MOLA_ICP_QPL=1 MOLA_ICP_BLOCKS_PER_CU=3" "MOLA_ICP_QPL=1 MOLA_ICP_BLOCKS_PER_CU=4" "MOLA_ICP_QPL=2 MOLA_ICP_BLOCKS_PER_CU=4"; do
    env $kv MOLA_ICP_COOP=0 timeout 300 python bench.py --n-local $1 --n-map $2 --cpu-baseline-iters 0 --dense-iters 0 --e2e 0 --batch-pairs 0 --shipped-iters 0 2>/dev/null | tail -n1 | line "$1x$2 $kv" || exit 1
  done
done
