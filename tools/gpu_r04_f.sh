#!/bin/bash
# round 4, step f: full GPU suite, the default bench, --force-dist over every collective, the c5 shard projection
set -o pipefail
mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -c 1500 $O/bench.err; exit 1; }
LEAN="--cpu-baseline-iters 0 --shipped-iters 0 --dense-iters 0 --e2e 0 --batch-pairs 0 --c5-map 0"
for ar in plain local rccl hook; do
  if [ $ar = plain ]; then X=""; else X="--force-dist --allreduce $ar"; fi
  timeout -k 10 300 python bench.py $LEAN $X > $O/bench_$ar.json 2> $O/bench_$ar.err || { tail -c 1500 $O/bench_$ar.err; exit 1; }
done
timeout -k 10 600 python tools/shard_step.py --config c5 > $O/shard_step_c5.jsonl 2> $O/shard_step_c5.err || { tail -c 1500 $O/shard_step_c5.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 2 $LEAN --c5-map 10000000 > $O/bench_2ranks_shared.json 2> $O/bench_2ranks_shared.err || { tail -c 1500 $O/bench_2ranks_shared.err; exit 1; }
python - <<'PY'
import json
O = "gpurun_out/r04f/"
d = json.loads(open(O + "bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "roofline", "shipped_point2plane_gn", "time_to_pose", "c5_sharded", "odometry_stream", "odometry_stream_10hz", "odometry_stream_small", "odometry_stream_small_10hz"):
    v = d.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("ms_per_scan", "pmc", "flop_view", "workload", "note")}
        if "roofline" in v:
            v["roofline"] = {a: b for a, b in v["roofline"].items() if a not in ("note", "pmc")}
    print(k, json.dumps(v)[:1100])
print("align_e2e", json.dumps({k: v["gpu"]["ms"] for k, v in d.get("align_e2e", {}).items()}))
for ar in ("plain", "local", "rccl", "hook"):
    j = json.loads(open(O + f"bench_{ar}.json").read().strip().splitlines()[-1])
    print(ar, j["ms_per_step"], j["config"]["parallelism"], j["config"].get("comm_nranks"))
for ln in open(O + "shard_step_c5.jsonl"):
    j = json.loads(ln)
    print("c5 world", j["world"], "slowest", round(j["step_ms_slowest_rank"], 4), "mean", round(j["step_ms_mean_rank"], 4), "speedup", round(j["projected_speedup_before_collective"], 2),
          "kept", [r["map_points_kept"] for r in j["ranks"]][:8])
j = json.loads(open(O + "bench_2ranks_shared.json").read().strip().splitlines()[-1])
print("2 ranks shared:", j["ms_per_step"], j["config"]["parallelism"], j["config"]["comm_nranks"], json.dumps(j["c5_sharded"])[:600])
PY
