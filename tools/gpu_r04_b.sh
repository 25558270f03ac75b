#!/bin/bash
# round 4: whole GPU suite on the current library, then the odometry numbers A/B against round 3's library in the same lease
set -u
TAG=${1:-r04d}; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/$TAG/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
LINES_SHOWN=3 bash tools/gpu_lib_ab_script.sh $TAG "timeout -k 10 200 python tools/odometry_ab.py" r03.so cur.so
timeout -k 10 300 bash tools/rocprof_odometry.sh > /dev/null && cp gpurun_out/prof_odometry/timeline.txt gpurun_out/$TAG/odometry_timeline.txt && head -12 gpurun_out/$TAG/odometry_timeline.txt && tail -1 gpurun_out/$TAG/odometry_timeline.txt
