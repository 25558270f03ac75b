#!/bin/bash
# late vs early reservation of the next entry in the persistent kernels (MOLA_ICP_EARLY_POP): headline, shipped pipeline, batch, wave lifetimes
O=gpurun_out/r02_pop; mkdir -p $O
line() { python -c "
import json,sys; d=json.load(sys.stdin); s=d.get('shipped_point2plane_gn') or {}; b=d.get('config3_batch') or {}
print('$1', 'value %.0f it/s  kernel %.1f us  pairs/query %.0f | shipped %s it/s kernel %s ms | batch %s' % (d['value'], d['roofline']['kernel_ms']*1e3, d['roofline']['flop_view']['pairs_evaluated_per_query'], s.get('value'), s.get('kernel_ms'), (b.get('gpu') or b).get('pairs_per_s')))"; }
for kv in "X=1" "MOLA_ICP_EARLY_POP=1" "MOLA_ICP_QPL=1" "MOLA_ICP_QPL=1 MOLA_ICP_EARLY_POP=1" "MOLA_ICP_NO_SPLIT=1"; do
  env $kv timeout 300 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --e2e 0 --batch-pairs 24 2>$O/err.log | tail -n1 | line "$kv" || exit 1
done
MOLA_ICP_DEBUG_STATS=2 timeout 200 python tools/prof_nn.py --kernel tiled --reps 3 2>&1 | grep -v amdgpu.ids | tail -16
