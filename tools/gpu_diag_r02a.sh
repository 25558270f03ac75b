#!/bin/bash
# round-2 diagnostics: GPU tests, per-phase cycles of the tiled matcher at odometry size, the point-to-plane pipeline over iteration counts
O=gpurun_out/r02_diag; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1
MOLA_ICP_DEBUG_STATS=1 timeout 200 python tools/prof_nn.py --kernel tiled --reps 3 --n 100000 --m 100000 > $O/dbg1_100k.log 2>&1
MOLA_ICP_DEBUG_STATS=2 timeout 200 python tools/prof_nn.py --kernel tiled --reps 3 --n 100000 --m 100000 > $O/dbg2_100k.log 2>&1
MOLA_ICP_DEBUG_STATS=2 timeout 200 python tools/prof_nn.py --kernel tiled --reps 3 > $O/dbg2_1m.log 2>&1
for it in 5 10 20 40; do timeout 200 python tools/prof_p2pl.py --iters $it >> $O/p2pl_iters.log 2>&1; done
tail -5 $O/pytest_gpu.log; cat $O/dbg1_100k.log $O/dbg2_100k.log $O/dbg2_1m.log $O/p2pl_iters.log | grep -v amdgpu.ids
