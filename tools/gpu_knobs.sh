#!/bin/bash
# k_nn_tiled at C3 under the tuning knobs (kernel time from the profiled repetition of bench.py)
for kv in "X=1" "MOLA_ICP_NO_SPLIT=1" "MOLA_ICP_BLOCKS_PER_CU=4 MOLA_ICP_QPL=2" "MOLA_ICP_BLOCKS_PER_CU=4 MOLA_ICP_QPL=2 MOLA_ICP_NO_SPLIT=1" "MOLA_ICP_BLOCKS_PER_CU=2 MOLA_ICP_QPL=2" "MOLA_ICP_QPL=1"; do
  env $kv python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 --e2e 0 --batch-pairs 0 2>/dev/null | tail -n1 | python -c "
import json,sys; d=json.load(sys.stdin); print('$kv', 'value %.0f it/s  kernel %.1f us  pairs/query %.0f' % (d['value'], d['roofline']['kernel_ms']*1e3, d['roofline']['flop_view']['pairs_evaluated_per_query']))"
done
