#!/bin/bash
# a library variant that differs in k_nn_q4's translation unit only: tools/build_q4_variant.sh <name> [-D...]  ->  mola-fe-lidar_amd/lib/variants/<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../mola-fe-lidar_amd/csrc"
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c q4_launch.hip -o /tmp/q4_launch_$name.o
objs=$(ls ../build/*.o | grep -v q4_launch)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -o ../lib/variants/$name.so $objs /tmp/q4_launch_$name.o -ldl
echo "built variants/$name.so"
