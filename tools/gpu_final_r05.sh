#!/bin/bash
# round 4: everything the committed profiles/r05 and the docs quote, in one lease (run from the repository root on the GPU box):
#   bash tools/gpu_final_r05.sh          then, in the repository:
#   python tools/pmc_record.py gpurun_out/prof_r05_headline profiles/r05
#   python tools/pmc_record.py gpurun_out/prof_r05_small profiles/r05 --kernel k_nn_coop
#   python tools/pmc_record_planes.py gpurun_out/prof_r05_planes_c3 profiles/r05 ; ... prof_r05_planes_120k ...
set -o pipefail
O=gpurun_out/final_r05; mkdir -p $O
if [ -z "$SKIP_SUITE" ]; then echo "== GPU suite"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && exit $rc; fi
if [ "${PART:-a}" = a ]; then
echo "== bench (all legs)"; (time timeout -k 10 900 python bench.py) > $O/bench.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -n1 $O/bench.log > $O/bench_line.json
LEAN="--cpu-baseline-iters 0 --e2e 0 --batch-pairs 0 --dense-iters 0 --shipped-iters 0"
echo "== bench --gpus 2 (self-launched ranks sharing this GPU; node-local all-reduce; c5 leg)"; timeout -k 10 600 python bench.py --gpus 2 $LEAN > $O/bench_gpus2.json 2> $O/bench_gpus2.err || { tail -20 $O/bench_gpus2.err; exit 1; }
echo "== one rank through every collective"
for ar in plain both local rccl hook; do
  if [ $ar = plain ]; then X=""; elif [ $ar = both ]; then X="--force-dist"; else X="--force-dist --allreduce $ar"; fi
  timeout -k 10 300 python bench.py $LEAN --c5-map 0 $X > $O/bench_one_rank_$ar.json 2> $O/bench_one_rank_$ar.err || { tail -20 $O/bench_one_rank_$ar.err; exit 1; }
done
echo "== shard steps"; timeout -k 10 600 python tools/shard_step.py --config c3 > $O/shard_step_c3.jsonl 2> $O/shard_step_c3.err; timeout -k 10 900 python tools/shard_step.py --config c5 > $O/shard_step_c5.jsonl 2> $O/shard_step_c5.err
fi
if [ "${PART:-a}" = b ]; then
echo "== bench.py under rocprofv3 --kernel-trace --stats"; bash tools/rocprof_bench.sh > $O/rocprof_bench.txt 2>&1; tail -8 $O/rocprof_bench.txt
echo "== headline counters"; bash tools/rocprof_headline.sh r05_headline > $O/rocprof_headline.log 2>&1; tail -5 $O/rocprof_headline.log
echo "== odometry-size matcher counters"; bash tools/rocprof_small.sh r05_small > $O/rocprof_small.log 2>&1; tail -5 $O/rocprof_small.log
echo "== plane matcher counters"; bash tools/rocprof_planes.sh r05_planes_c3 1000000 > $O/planes_c3.log 2>&1; bash tools/rocprof_planes.sh r05_planes_120k 120000 > $O/planes_120k.log 2>&1; tail -3 $O/planes_120k.log
echo "== config 0 timeline"; bash tools/rocprof_config0.sh > $O/config0_kernel_trace.txt 2>&1; tail -3 $O/config0_kernel_trace.txt
echo "== odometry stream timeline"; bash tools/rocprof_odometry.sh > $O/rocprof_odometry.log 2>&1; tail -3 gpurun_out/prof_odometry/timeline.txt
echo "== headline iteration timeline"; bash tools/rocprof_timeline_any.sh 1000000 p2p 12 > $O/c3_timeline.txt 2>&1; tail -6 $O/c3_timeline.txt
fi
[ "${PART:-a}" = a ] && python - <<'PY'
import json
O = "gpurun_out/final_r05/"
d = json.loads(open(O + "bench_line.json").read())
for k in ("value", "ms_per_step", "shipped_point2plane_gn", "time_to_pose", "cold_start", "c5_sharded", "odometry_stream", "odometry_stream_10hz", "odometry_stream_small", "odometry_stream_small_10hz", "config3_batch", "config3_batch_shipped", "loop_closure_montecarlo"):
    v = d.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("ms_per_scan", "pmc", "flop_view", "workload", "note", "roofline", "cpu")}
    print(k, json.dumps(v)[:700])
print("roofline", json.dumps({a: b for a, b in d["roofline"].items() if a not in ("flop_view", "pmc")})[:600])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:400])
print("align_e2e", json.dumps({k: v["gpu"]["ms"] for k, v in d.get("align_e2e", {}).items()}))
for ar in ("plain", "both", "local", "rccl", "hook"):
    j = json.loads(open(O + f"bench_one_rank_{ar}.json").read().strip().splitlines()[-1])
    print(ar, j["ms_per_step"], j["ms_per_step_repeat_on_warm_state"], j["config"]["parallelism"], j["config"].get("comm_nranks"), json.dumps(j["config"].get("allreduce")))
j = json.loads(open(O + "bench_gpus2.json").read().strip().splitlines()[-1])
print("2 ranks shared:", j["ms_per_step"], j["config"]["parallelism"], j["config"]["comm_nranks"], json.dumps(j["config"]["shard_balance"])[:300], json.dumps({a: b for a, b in j["c5_sharded"].items() if a != "workload"})[:600])
PY
