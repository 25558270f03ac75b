#!/bin/bash
# shipped pipeline under knobs, alternating on one box
for r in 1 2; do
for kv in "X=1" "MOLA_ICP_BLOCKS_PER_CU=3" "MOLA_ICP_BLOCKS_PER_CU=5"; do
  env $kv python bench.py --cpu-baseline-iters 0 --dense-iters 0 --e2e 0 --batch-pairs 0 2>/dev/null | tail -n1 | python -c "
import json,sys; d=json.load(sys.stdin); s=d['shipped_point2plane_gn']; print('round $r $kv shipped %.0f it/s  kernel %.1f us' % (s['value'], s['kernel_ms']*1e3))"
done; done
python tools/prof_config0.py 2>&1 | grep "profiling=False"
