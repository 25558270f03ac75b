#!/bin/bash
# round 4: the hand-written prepare chain -- its own tests, (full: the whole GPU suite,) the odometry stream plain and under rocprofv3
#   bash tools/gpu_r04_a.sh [quick|full] [tag]
set -u
MODE=${1:-full}; TAG=${2:-r04a}
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests/test_gpu_prepare.py -x -q -m gpu > gpurun_out/$TAG/prepare_tests.log 2>&1; rc=$?
tail -3 gpurun_out/$TAG/prepare_tests.log
[ $rc -ne 0 ] && exit $rc
if [ "$MODE" == "full" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1; rc=$?
  tail -3 gpurun_out/$TAG/gpu_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python tools/prof_odometry_stream.py 24 > gpurun_out/$TAG/odometry_stream.txt 2>&1 && cut -c1-330 gpurun_out/$TAG/odometry_stream.txt | tail -2
timeout -k 10 300 bash tools/rocprof_odometry.sh > /dev/null && cp gpurun_out/prof_odometry/timeline.txt gpurun_out/$TAG/odometry_timeline.txt && cat gpurun_out/$TAG/odometry_timeline.txt
