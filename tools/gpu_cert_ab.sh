#!/bin/bash
# certified neighbour lists of the point-to-plane matcher on / off: shipped pipeline at C3 (20 fixed iterations), e2e legs
O=gpurun_out/r02_cert; mkdir -p $O
for kv in "X=1" "MOLA_ICP_NO_CERTIFY=1"; do
  env $kv timeout 300 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --batch-pairs 0 2>$O/err.log | tail -n1 > $O/line.json || exit 1
  python - "$kv" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r02_cert/line.json'))
e=d.get('align_e2e',{}); s=d.get('shipped_point2plane_gn') or {}
print(sys.argv[1], 'value %.0f it/s  kernel %.1f us' % (d['value'], d['roofline']['kernel_ms']*1e3), '| shipped %.0f it/s kernel %.3f ms' % (s.get('value',0), s.get('kernel_ms',0)),
      '| e2e', {k:(round(v['gpu']['ms'],3), v['gpu']['iterations']) for k,v in e.items()})
PY
done
