mkdir -p gpurun_out; : > gpurun_out/exp.txt
(timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3) >> gpurun_out/exp.txt
timeout 300 python tools/shard_step.py 2>&1 | grep "z-order" >> gpurun_out/exp.txt
for q in 1 2; do echo "== QPL=$q batch" >> gpurun_out/exp.txt; MOLA_ICP_QPL=$q timeout 300 python tools/bench_batch.py --pairs 8 2>&1 | tail -3 >> gpurun_out/exp.txt; done
echo "== QPL=1 at 1M" >> gpurun_out/exp.txt
MOLA_ICP_QPL=1 timeout 200 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))" >> gpurun_out/exp.txt 2>&1
