#!/bin/bash
# a library variant that differs in ONE translation unit only (k_nn_q4's by default; TU=knn_q4_launch: k_knn_q4's):
#   [TU=knn_q4_launch] tools/build_q4_variant.sh <name> [-D...]  ->  mola-fe-lidar_amd/lib/variants/<name>.so
set -e
name=$1; shift
TU=${TU:-q4_launch}
cd "$(dirname "$0")/../mola-fe-lidar_amd/csrc"
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c $TU.hip -o /tmp/${TU}_$name.o
objs=$(ls ../build/*.o | grep -v "/$TU.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -o ../lib/variants/$name.so $objs /tmp/${TU}_$name.o -ldl
echo "built variants/$name.so"
