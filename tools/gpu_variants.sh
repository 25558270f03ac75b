mkdir -p gpurun_out
cp mola-fe-lidar_amd/lib/libmola_icp_amd.so /tmp/lib_orig.so
: > gpurun_out/variants.txt
cp build/variants/lib_dyn.so mola-fe-lidar_amd/lib/libmola_icp_amd.so
for n in 500000 1000000; do
  echo "== n=$n" >> gpurun_out/variants.txt
  timeout 25 python tools/prof_p2pl.py --n $n --iters 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/variants.txt; echo "rc=$?" >> gpurun_out/variants.txt
done
cp /tmp/lib_orig.so mola-fe-lidar_amd/lib/libmola_icp_amd.so
