#!/bin/bash
# the whole GPU suite (or the files given), log under gpurun_out/<tag>/
set -u
TAG=${1:-suite}; shift || true
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest ${@:-tests} -x -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1; rc=$?
tail -4 gpurun_out/$TAG/gpu_tests.log
exit $rc
