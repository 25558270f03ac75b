#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py ITSELF (the driver's command, headline leg only): the kernel's average duration in
# the stats must agree with the `roofline.kernel_ms` the same run prints.   bash tools/rocprof_bench.sh   (on the GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_bench; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --e2e 0 --batch-pairs 0 --dense-iters 0 --shipped-iters 0 > $OUT/bench.json 2> $OUT/bench.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<PY
import csv, json
j = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
print("bench.py: value %.0f it/s, ms_per_step %.4f, roofline.kernel_ms %.4f (HIP events, this run)" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
for r in rows[:6]:
    print("%-60s calls %6s  avg %9.1f ns  total %5.1f %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
