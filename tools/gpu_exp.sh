mkdir -p gpurun_out; : > gpurun_out/exp.txt
run() { echo "== $*" >> gpurun_out/exp.txt; env "$@" timeout 200 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['pose_err_vs_gt'])" >> gpurun_out/exp.txt 2>&1; }
for b in 512 1024 2048; do run MOLA_ICP_ACC_BLOCKS=$b; done
for s in "MOLA_ICP_SPLIT2=0 MOLA_ICP_SPLIT4=1e30" "MOLA_ICP_SPLIT2=0 MOLA_ICP_SPLIT4=0" "A=1"; do echo "== $s" >> gpurun_out/exp.txt; env $s timeout 300 python tools/bench_batch.py --pairs 8 2>&1 | tail -3 >> gpurun_out/exp.txt; done
