#!/bin/bash
set -o pipefail
O=gpurun_out/r04i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sharded_gpu.py tests/test_bench_launch.py -m gpu -x -q > $O/pytest.log 2>&1
rc=$?; tail -5 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python tools/shard_step.py --config c5 > $O/shard_step_c5.jsonl 2> $O/shard_step_c5.err || { tail -c 1500 $O/shard_step_c5.err; exit 1; }
python - <<'PY'
import json
for ln in open("gpurun_out/r04i/shard_step_c5.jsonl"):
    j = json.loads(ln)
    print("c5 world", j["world"], j["cuts"], "slowest", round(j["step_ms_slowest_rank"], 4), "mean", round(j["step_ms_mean_rank"], 4), "speedup", round(j["projected_speedup_before_collective"], 2),
          "ms", [round(r["ms_per_iteration"], 3) for r in j["ranks"]], "q", [r["queries"] for r in j["ranks"]])
PY
