#!/bin/bash
# A/B of environment knobs on ONE box, alternating, N rounds: bash tools/gpu_ab.sh "A=1" "MOLA_ICP_NO_SPLIT=1" ...
R=${ROUNDS:-3}
for r in $(seq 1 $R); do
  for kv in "$@"; do
    env $kv python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 --e2e 0 --batch-pairs 0 2>/dev/null | tail -n1 | python -c "
import json,sys; d=json.load(sys.stdin); print('round $r', '$kv', 'value %.0f it/s  kernel %.1f us  pairs/query %.0f' % (d['value'], d['roofline']['kernel_ms']*1e3, d['roofline']['flop_view']['pairs_evaluated_per_query']))"
  done
done
