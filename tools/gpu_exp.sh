mkdir -p gpurun_out; : > gpurun_out/exp.txt
(timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3) >> gpurun_out/exp.txt
timeout 60 python tools/prof_p2pl.py --n 1000000 --iters 10 2>&1 | grep -v amdgpu.ids | cut -c1-110 >> gpurun_out/exp.txt
timeout 60 python tools/prof_p2pl.py --n 100000 --iters 10 2>&1 | grep -v amdgpu.ids | cut -c1-110 >> gpurun_out/exp.txt
timeout 200 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), (d.get('shipped_point2plane_gn') or {}).get('value'))" >> gpurun_out/exp.txt 2>&1
