#!/bin/bash
set -o pipefail
O=gpurun_out/r04k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pose_sequence.py tests/test_point2plane.py tests/test_gpu_batch_planes.py -m gpu -x -q > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/odometry_ab.py > $O/odometry.txt 2> $O/odometry.err || { tail -c 1000 $O/odometry.err; exit 1; }
cat $O/odometry.txt
timeout -k 10 300 bash tools/rocprof_odometry.sh > $O/rocprof.log 2>&1 || { tail -c 1000 $O/rocprof.log; exit 1; }
cat gpurun_out/prof_odometry/timeline.txt
