#!/bin/bash
# round 4: plane-matcher changes -- the whole GPU suite, certificate statistics, same-lease A/B against the previous library, timeline
set -u
TAG=${1:-r04e}; mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/$TAG/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/dbg_cert.py 2> gpurun_out/$TAG/dbg_cert.txt; grep -A4 "scan 15" gpurun_out/$TAG/dbg_cert.txt | cut -c1-250
LINES_SHOWN=3 bash tools/gpu_lib_ab_script.sh $TAG "timeout -k 10 200 python tools/odometry_ab.py" prev.so cur.so
timeout -k 10 300 bash tools/rocprof_odometry.sh > /dev/null && cp gpurun_out/prof_odometry/timeline.txt gpurun_out/$TAG/odometry_timeline.txt && cat gpurun_out/$TAG/odometry_timeline.txt
