#!/bin/bash
# the shipped point-to-plane pipeline at C3, R rounds: value + matcher time per launch
R=${ROUNDS:-3}
for r in $(seq 1 $R); do
  python bench.py --cpu-baseline-iters 0 --dense-iters 0 --e2e 0 --batch-pairs 0 2>/dev/null | tail -n1 | python -c "
import json,sys; d=json.load(sys.stdin); s=d['shipped_point2plane_gn']; print('round $r shipped %.0f it/s  kernel %.1f us | p2p %.0f it/s kernel %.1f us' % (s['value'], s['kernel_ms']*1e3, d['value'], d['roofline']['kernel_ms']*1e3))"
done
for it in 1 5 20; do python tools/prof_p2pl.py --iters $it 2>/dev/null | tail -1; done
