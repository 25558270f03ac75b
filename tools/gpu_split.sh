#!/bin/bash
# heavy-item splitting at C3: parity tests of the tiled kernels, then bench with / without the split
O=gpurun_out/${1:-r02_split}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_liveness.py -m gpu -x -q) > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for kv in "X=1" "MOLA_ICP_NO_SPLIT=1"; do
  env $kv python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 --e2e 0 --batch-pairs 0 2>/dev/null | tail -n1 | python -c "
import json,sys; d=json.load(sys.stdin); print('$kv', 'value %.0f it/s  kernel %.1f us  pairs/query %.0f' % (d['value'], d['roofline']['kernel_ms']*1e3, d['roofline']['flop_view']['pairs_evaluated_per_query']))"
done
MOLA_ICP_DEBUG_STATS=2 python tools/prof_nn.py --kernel tiled --reps 12 2>&1 | grep "last tiled launch\|per XCD" | tail -2
