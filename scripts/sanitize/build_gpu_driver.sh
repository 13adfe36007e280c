#!/bin/bash
# Builds variants/san/gpu_host_san: gpu_host_san.cpp + the scene library + librt_amd's sources in one executable,
# host code instrumented with ASan + UBSan (device code is not).  Build here, run on the GPU box:
#   ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 variants/san/gpu_host_san
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
O=$R/variants/san
mkdir -p $O
CL=/opt/rocm/lib/llvm/bin/clang++
HOSTSAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
DEV="-std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DRT_BVH_WIDTH=4"
hipcc -O1 -g $DEV $HOSTSAN -c $R/raytracing_amd/csrc/rt_api.hip -o $O/rt_api.o
hipcc -O3 $DEV $HOSTSAN -c $R/raytracing_amd/csrc/rt_kernels.hip -o $O/rt_kernels.o
for f in $R/scripts/sanitize/gpu_host_san.cpp $R/raytracing_amd/host/{raytracing,scenes,capi,image_io}.cpp $R/raytracing_amd/csrc/rt_bvh.cpp; do
  $CL -O1 -g -std=c++17 -ffp-contract=off -DRT_BVH_WIDTH=4 -fsanitize=address,undefined -fno-omit-frame-pointer -c $f -o $O/$(basename $f .cpp).o
done
hipcc --offload-arch=gfx950 -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize -o $O/gpu_host_san $O/*.o -lpthread
rm -f $O/*.o
echo built $O/gpu_host_san
