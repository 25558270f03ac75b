#!/bin/bash
# as gpu_lib_ab.sh, for any script that prints its own summary: tools/gpu_lib_ab_script.sh <tag> "<command>" a.so b.so ...  (first variant run twice)
tag=$1; cmd=$2; shift 2
L=mola-fe-lidar_amd/lib; mkdir -p gpurun_out/$tag; cp $L/libmola_icp_amd.so /tmp/orig_lib.so
for v in "$@" "$1"; do
  n=$(basename $v .so); [ -f gpurun_out/$tag/$n.txt ] && n=${n}_again
  cp $L/variants/$(basename $v) $L/libmola_icp_amd.so
  bash -c "$cmd" > gpurun_out/$tag/$n.txt 2>&1 || { tail -5 gpurun_out/$tag/$n.txt; cp /tmp/orig_lib.so $L/libmola_icp_amd.so; exit 1; }
  echo "== $n"; grep -v "^W2\|^E2\|amdgpu.ids" gpurun_out/$tag/$n.txt | head -${LINES_SHOWN:-4}
done
cp /tmp/orig_lib.so $L/libmola_icp_amd.so
