#!/bin/bash
# k_nn_q4 built with different launch bounds (mola-fe-lidar_amd/lib/variants/q4_wg<n>.so), same lease: tools/gpu_q4_variants.sh <tag> [sizes...]
tag=${1:-q4v}; shift
mkdir -p gpurun_out/$tag
for v in mola-fe-lidar_amd/lib/variants/q4_wg*.so; do
  echo "== $(basename $v)" >> gpurun_out/$tag/variants.txt
  MOLA_ICP_LIB_PATH=$PWD/$v timeout -k 10 200 python tools/q4_ab.py "$@" >> gpurun_out/$tag/variants.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/$tag/variants.txt
