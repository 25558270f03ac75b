#!/bin/bash
set -o pipefail
O=gpurun_out/r04m2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pose_sequence.py tests/test_point2plane.py tests/test_front_end.py -m gpu -x -q > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
bash tools/kernel_times.sh k_bootstrap,k_quality > $O/kernel_times.txt 2>&1; cat $O/kernel_times.txt | grep -v amdgpu.ids
timeout -k 10 300 python tools/odometry_ab.py > $O/odometry.txt 2> $O/odometry.err || { tail -c 1000 $O/odometry.err; exit 1; }
cat $O/odometry.txt
