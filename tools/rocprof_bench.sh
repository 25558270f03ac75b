#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py ITSELF (the driver's command, headline leg only): the kernel's average duration in
# the stats must agree with the `roofline.kernel_ms` the same run prints.   bash tools/rocprof_bench.sh   (on the GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_bench; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline-iters 0 --e2e 0 --batch-pairs 0 --dense-iters 0 --shipped-iters 0 --c5-map 0 > $OUT/bench.json 2> $OUT/bench.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - <<PY
import csv, json
j = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
print("bench.py: value %.0f it/s, ms_per_step %.4f, roofline.kernel_ms %.4f (HIP events, this run)" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"]))
for r in rows[:6]:
    print("%-60s calls %6s  avg %9.1f ns  total %5.1f %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
# the dominant kernel's launches one by one: a mean that holds a few cold / unseeded launches is not the figure to put beside the
# line's median of HIP-event durations
import statistics
dur = {}
for r in csv.DictReader(open("$OUT/kernel_trace.csv")):
    dur.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
top = max(dur, key=lambda k: sum(dur[k]))
v = sorted(dur[top])
print("%s: %d launches, us: median %.1f  mean %.1f  p5 %.1f  p95 %.1f  max %.1f" % (top[-50:], len(v), statistics.median(v), sum(v) / len(v), v[len(v) // 20], v[len(v) * 19 // 20], v[-1]))
json.dump({"kernel": top, "launches": len(v), "median_us": statistics.median(v), "mean_us": sum(v) / len(v), "p95_us": v[len(v) * 19 // 20], "max_us": v[-1],
           "bench_line": {k: j[k] for k in ("value", "ms_per_step", "value_repeat_on_warm_state") if k in j}, "roofline_kernel_ms": j["roofline"]["kernel_ms"]},
          open("$OUT/bench_py_under_rocprof.json", "w"), indent=1)
PY
