#!/bin/bash
# a development step on the GPU box: some test files, then the odometry A/B numbers and the timeline of one scan
#   bash tools/gpu_step.sh <tag> "<pytest files>"
set -o pipefail
TAG=${1:-step}; FILES=${2:-tests}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest $FILES -m gpu -x -q > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/odometry_ab.py > $O/odometry.txt 2> $O/odometry.err || { tail -c 1000 $O/odometry.err; exit 1; }
cat $O/odometry.txt
timeout -k 10 300 bash tools/rocprof_odometry.sh > $O/rocprof.log 2>&1 || { tail -c 1000 $O/rocprof.log; exit 1; }
cat gpurun_out/prof_odometry/timeline.txt
