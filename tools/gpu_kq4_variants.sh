#!/bin/bash
# same-lease A/B of library variants that differ in k_knn_q4's translation unit (tools/build_q4_variant.sh with TU=knn_q4_launch):
#   bash tools/gpu_kq4_variants.sh <out tag> <variant> [<variant> ...]     ("product" = the library as built)
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2; do
  for v in "$@"; do
    if [ $v = product ]; then unset MOLA_ICP_LIB_PATH; else export MOLA_ICP_LIB_PATH=$PWD/mola-fe-lidar_amd/lib/variants/$v.so; fi
    echo "== $v (round $round)"
    MOLA_ICP_KNN_Q4=${KNN_Q4:-1} timeout -k 10 200 python tools/odometry_ab.py 2>&1 | grep -v amdgpu.ids || exit 1
    MOLA_ICP_KNN_Q4=${KNN_Q4:-1} timeout -k 10 100 python tools/prof_mc.py 2>&1 | grep -v amdgpu.ids | tail -1
  done
done > $O/variants.txt 2>&1
cat $O/variants.txt
