mkdir -p gpurun_out; : > gpurun_out/exp.txt
run() { echo "== $*" >> gpurun_out/exp.txt; env "$@" timeout 200 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))" >> gpurun_out/exp.txt 2>&1; }
run MOLA_ICP_QPL=1 MOLA_ICP_BLOCKS_PER_CU=3
run MOLA_ICP_QPL=1 MOLA_ICP_BLOCKS_PER_CU=4
run MOLA_ICP_QPL=1 MOLA_ICP_BLOCKS_PER_CU=5
run MOLA_ICP_QPL=2 MOLA_ICP_BLOCKS_PER_CU=3
