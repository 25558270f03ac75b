mkdir -p gpurun_out
(time timeout 700 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
timeout 300 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 > gpurun_out/bench_tiled.log 2>&1
MOLA_ICP_DEBUG_STATS=1 timeout 200 python tools/prof_nn.py --kernel tiled --reps 3 > gpurun_out/dbg.log 2>&1
timeout 400 python tools/bench_batch.py --pairs 16 > gpurun_out/batch.log 2>&1
for b in 2 4; do MOLA_ICP_BLOCKS_PER_CU=$b timeout 300 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 > gpurun_out/bench_tiled_b$b.log 2>&1; done
