#!/bin/bash
# Builds variants/librt_amd_<name>.so: the GPU library with extra compiler flags (development aid for A/B runs through
# RT_AMD_LIBRARY).  usage: scripts/build_variant.sh <name> [-DRT_...=... ...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/raytracing_amd/csrc
obj=/tmp/rt_variant_$name
mkdir -p "$obj" "$root/variants"
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DRT_BVH_WIDTH=4 -DRT_DEV_KNOBS $*"
hipcc $flags -c "$src/rt_kernels.hip" -o "$obj/rt_kernels.o" &
hipcc $flags -c "$src/rt_api.hip" -o "$obj/rt_api.o" &
hipcc -O3 -std=c++17 -fPIC -DRT_BVH_WIDTH=4 -c "$src/rt_bvh.cpp" -o "$obj/rt_bvh.o" &
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o "$root/variants/librt_amd_$name.so" "$obj/rt_api.o" "$obj/rt_kernels.o" "$obj/rt_bvh.o" -lpthread
echo "built variants/librt_amd_$name.so"
