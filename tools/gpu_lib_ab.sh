#!/bin/bash
# A/B of whole library builds in one lease: tools/gpu_lib_ab.sh <tag> <variant.so> [<variant.so> ...] -- [bench args]
# (variants are built beforehand into mola-fe-lidar_amd/lib/variants/, which travels with the snapshot; the first is run twice)
tag=$1; shift
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" == "--" ] && shift
L=mola-fe-lidar_amd/lib; mkdir -p gpurun_out/$tag; cp $L/libmola_icp_amd.so /tmp/orig_lib.so
Q="--cpu-baseline-iters 0 --e2e 0 --batch-pairs 0 --dense-iters 0"
for v in "${libs[@]}" "${libs[0]}"; do
  n=$(basename $v .so); [ -f gpurun_out/$tag/$n.json ] && n=${n}_again
  cp $L/variants/$(basename $v) $L/libmola_icp_amd.so
  python bench.py $Q "$@" > gpurun_out/$tag/$n.json 2> gpurun_out/$tag/$n.err || { tail -5 gpurun_out/$tag/$n.err; cp /tmp/orig_lib.so $L/libmola_icp_amd.so; exit 1; }
  python - <<PY
import json
j = json.loads(open("gpurun_out/$tag/$n.json").read().strip().splitlines()[-1])
s = j.get("shipped_point2plane_gn", {})
print("%-12s it/s %.0f  ms/step %.4f  matcher ms %.4f  pairs/query %s  shipped it/s %.0f" % ("$n", j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"].get("pairs_evaluated_per_query"), s.get("value", 0)))
PY
done
cp /tmp/orig_lib.so $L/libmola_icp_amd.so
