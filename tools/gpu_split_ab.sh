#!/bin/bash
# k_nn_tiled at C3 under the knob settings given as arguments (headline bench only)
line() { python -c "
import json,sys; d=json.load(sys.stdin)
print('$1', 'value %.0f it/s  kernel %.1f us  pairs/query %.0f' % (d['value'], d['roofline']['kernel_ms']*1e3, d['roofline']['flop_view']['pairs_evaluated_per_query']))"; }
for kv in "$@"; do
  env $kv timeout 300 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --e2e 0 --batch-pairs 0 --shipped-iters 0 2>/dev/null | tail -n1 | line "$kv" || exit 1
done
