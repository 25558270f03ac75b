mkdir -p gpurun_out; : > gpurun_out/exp.txt
(timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3) >> gpurun_out/exp.txt
run() { echo "== $*" >> gpurun_out/exp.txt; timeout 200 python bench.py "$@" --cpu-baseline-iters 0 --dense-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['config']['parallelism'], (d.get('shipped_point2plane_gn') or {}).get('value'))" >> gpurun_out/exp.txt 2>&1; }
run --shipped-iters 20
run --shipped-iters 0 --force-dist
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-gpu --steps 20 --warmup 3 --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks share-gpu', round(d['value'],1), d['pose_err_vs_gt'])" >> gpurun_out/exp.txt 2>&1
