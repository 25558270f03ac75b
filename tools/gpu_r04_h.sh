#!/bin/bash
set -o pipefail
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pose_sequence.py tests/test_point2plane.py tests/test_front_end.py tests/test_staged_pipeline.py tests/test_sharded_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
rc=$?; tail -5 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
for v in on off; do
  if [ $v = off ]; then export MOLA_ICP_NO_QUALITY_LISTS=1; else unset MOLA_ICP_NO_QUALITY_LISTS; fi
  timeout -k 10 300 python tools/odometry_ab.py > $O/odometry_$v.txt 2> $O/odometry_$v.err || { tail -c 1000 $O/odometry_$v.err; exit 1; }
  echo "quality from lists $v:"; tail -4 $O/odometry_$v.txt
done
