mkdir -p gpurun_out; : > gpurun_out/exp.txt
for n in 1000000 500000 250000 125000; do
echo "== n_local $n" >> gpurun_out/exp.txt
timeout 200 python bench.py --n-local $n --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))" >> gpurun_out/exp.txt 2>&1
done
echo "== force-dist rccl 1 rank, n_local 125000" >> gpurun_out/exp.txt
timeout 200 python bench.py --n-local 125000 --force-dist --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))" >> gpurun_out/exp.txt 2>&1
