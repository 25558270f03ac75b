#!/bin/bash
# the whole bench line (all legs) + rocprofv3 passes of the headline and the odometry-size matcher
O=gpurun_out/${1:-r02_bench}; mkdir -p $O
(time timeout 900 python bench.py) > $O/bench.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -n1 $O/bench.log | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps({k: d[k] for k in d if k not in ('roofline',)}, indent=0)[:6000]); print(json.dumps(d['roofline'])[:1500])"
grep real $O/bench.err
bash tools/rocprof_headline.sh ${1:-r02_bench}_headline > $O/rocprof_headline.log 2>&1; tail -30 $O/rocprof_headline.log
bash tools/rocprof_small.sh ${1:-r02_bench}_small > $O/rocprof_small.log 2>&1; tail -34 $O/rocprof_small.log
