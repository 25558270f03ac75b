#!/bin/bash
# round-2 check after a kernel change: GPU tests, the odometry-size / batch numbers, the headline bench line
O=gpurun_out/${1:-r02_cycle}; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
timeout 600 python tools/bench_configs.py > $O/configs.log 2>&1
timeout 300 python bench.py --cpu-baseline-iters 0 --dense-iters 0 > $O/bench.log 2>&1
MOLA_ICP_DEBUG_STATS=2 timeout 200 python tools/prof_nn.py --kernel tiled --reps 3 --n 100000 --m 100000 > $O/dbg2_100k.log 2>&1
tail -5 $O/pytest_gpu.log; grep -v amdgpu.ids $O/configs.log; tail -c 1500 $O/bench.log; grep -v amdgpu.ids $O/dbg2_100k.log
