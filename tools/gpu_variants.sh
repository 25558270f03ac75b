mkdir -p gpurun_out
cp mola-fe-lidar_amd/lib/libmola_icp_amd.so /tmp/lib_orig.so
: > gpurun_out/variants.txt
for f in build/variants/lib_*.so; do
  cp $f mola-fe-lidar_amd/lib/libmola_icp_amd.so
  echo "== $f" >> gpurun_out/variants.txt
  timeout 40 python tools/prof_p2pl.py --n 1000000 --iters 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/variants.txt; echo "rc=$?" >> gpurun_out/variants.txt
done
cp /tmp/lib_orig.so mola-fe-lidar_amd/lib/libmola_icp_amd.so
