#!/bin/bash
set -o pipefail
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 300 python tools/odometry_small_probe.py > $O/small_probe.txt 2> $O/small_probe.err || { tail -c 1500 $O/small_probe.err; exit 1; }
cat $O/small_probe.txt
MOLA_ICP_DEBUG_STATS=1 timeout -k 10 300 python tools/shard_step.py --config c5 --worlds 1 --iters 6 > $O/c5_dbg.jsonl 2> $O/c5_dbg.err || { tail -c 1500 $O/c5_dbg.err; exit 1; }
grep "mola_icp debug" $O/c5_dbg.err | tail -24
LEAN="--cpu-baseline-iters 0 --shipped-iters 0 --dense-iters 0 --e2e 0 --batch-pairs 0 --c5-map 0"
for ar in plain local rccl; do
  if [ $ar = plain ]; then X=""; else X="--force-dist --allreduce $ar"; fi
  timeout -k 10 300 python bench.py $LEAN $X > $O/bench_$ar.json 2> $O/bench_$ar.err || { tail -c 1500 $O/bench_$ar.err; exit 1; }
  python -c "
import json,sys
j=json.loads(open('$O/bench_$ar.json').read().strip().splitlines()[-1]); print('$ar', j['ms_per_step'], j['ms_per_step_before_closing_barrier'], j['config']['parallelism'])"
done
