#!/bin/bash
# round 3: A/B of one environment knob on the headline bench (no CPU legs):  tools/gpu_r3_ab.sh <tag> <KNOB=1> [more bench args]
tag=$1; knob=$2; shift 2
mkdir -p gpurun_out/$tag
Q="--cpu-baseline-iters 0 --e2e 0 --batch-pairs 0 --dense-iters 0"
python bench.py $Q "$@" > gpurun_out/$tag/default.json 2> gpurun_out/$tag/default.err || exit 1
env $knob python bench.py $Q "$@" > gpurun_out/$tag/knob.json 2> gpurun_out/$tag/knob.err || exit 1
python bench.py $Q "$@" > gpurun_out/$tag/default2.json 2> gpurun_out/$tag/default2.err || exit 1
python - <<PY
import json
for n in ("default", "knob", "default2"):
    j = json.loads(open("gpurun_out/$tag/%s.json" % n).read().strip().splitlines()[-1])  # (RCCL prints a banner first)
    s = j.get("shipped_point2plane_gn", {})
    print(n, "it/s %.0f  ms/step %.4f  matcher ms %.4f  shipped it/s %.0f" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], s.get("value", 0)))
PY
