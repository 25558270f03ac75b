mkdir -p gpurun_out
MOLA_ICP_DEBUG_STATS=1 timeout 200 python tools/prof_nn.py --kernel tiled --reps 3 > gpurun_out/dbg.log 2>&1
MOLA_ICP_DEBUG_STATS=1 timeout 200 python tools/prof_nn.py --kernel tiled --reps 3 --n 100000 --m 100000 > gpurun_out/dbg100k.log 2>&1
