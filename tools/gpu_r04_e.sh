#!/bin/bash
# round 4, step e: the full GPU suite on the current tree, then the bench with the paced / small-cloud legs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04e_pytest.log 2>&1
rc=$?
tail -5 gpurun_out/r04e_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r04e_bench.json 2> gpurun_out/r04e_bench.err
rc=$?
tail -c 600 gpurun_out/r04e_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04e_bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "roofline", "shipped_point2plane_gn", "time_to_pose", "odometry_stream", "odometry_stream_10hz", "odometry_stream_small", "odometry_stream_small_10hz"):
    v = d.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("ms_per_scan", "pmc")}
    print(k, json.dumps(v)[:1400])
print("align_e2e", json.dumps({k: v["gpu"] for k, v in d.get("align_e2e", {}).items()})[:1500])
PY
exit $rc
