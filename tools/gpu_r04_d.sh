#!/bin/bash
# round 4: quick same-lease A/B of two library builds on the odometry numbers (+ the sequence tests of the current one, + the timeline)
set -u
TAG=${1:-r04g}; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_pose_sequence.py tests/test_point2plane.py tests/test_gpu_batch_planes.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/$TAG/seq_tests.log 2>&1; rc=$?
tail -3 gpurun_out/$TAG/seq_tests.log
[ $rc -ne 0 ] && exit $rc
LINES_SHOWN=3 bash tools/gpu_lib_ab_script.sh $TAG "timeout -k 10 200 python tools/odometry_ab.py" prev.so cur.so
timeout -k 10 300 bash tools/rocprof_odometry.sh > /dev/null && cp gpurun_out/prof_odometry/timeline.txt gpurun_out/$TAG/odometry_timeline.txt && sed -n 8,30p gpurun_out/$TAG/odometry_timeline.txt
