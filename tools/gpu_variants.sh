# runs bench.py (tiled only) for each prebuilt variant in build/variants/ at 2 and 3 blocks per CU
mkdir -p gpurun_out
cp mola-fe-lidar_amd/lib/libmola_icp_amd.so /tmp/lib_orig.so
: > gpurun_out/variants.txt
for f in build/variants/lib_*.so; do
  cp $f mola-fe-lidar_amd/lib/libmola_icp_amd.so
  for b in 2 3; do
    MOLA_ICP_BLOCKS_PER_CU=$b timeout 300 python bench.py --cpu-baseline-iters 0 --dense-iters 0 --shipped-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', $b, round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))" >> gpurun_out/variants.txt 2>&1
  done
  timeout 300 python tools/bench_batch.py --pairs 8 2>&1 | grep sequential | sed "s|^|$f |" >> gpurun_out/variants.txt
done
cp /tmp/lib_orig.so mola-fe-lidar_amd/lib/libmola_icp_amd.so
