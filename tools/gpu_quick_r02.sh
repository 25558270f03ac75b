#!/bin/bash
# quick look after a cooperative-kernel change: parity tests of the small/batched path, odometry-size numbers, per-phase diagnostics
O=gpurun_out/${1:-r02_quick}; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "coop or batch or multi_init or montecarlo or config2 or tiled or align" ) > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
timeout 600 python tools/bench_configs.py --skip-1m > $O/configs.log 2>&1
MOLA_ICP_DEBUG_STATS=2 timeout 200 python tools/prof_nn.py --kernel tiled --reps 3 --n 100000 --m 100000 > $O/dbg2_100k.log 2>&1
tail -3 $O/pytest_gpu.log; grep -v amdgpu.ids $O/configs.log | cut -c1-400; grep -v amdgpu.ids $O/dbg2_100k.log | tail -6
