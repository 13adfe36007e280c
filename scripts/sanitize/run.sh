#!/bin/bash
# Host code under AddressSanitizer + UndefinedBehaviorSanitizer (and the threaded BVH build under ThreadSanitizer):
# the BVH builder on ordinary and degenerate inputs, the scene library building and flattening every scene,
# the CPU oracle rendering every scene with both random number generators.  No GPU needed.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
T=$(mktemp -d)
SAN="-O1 -g -std=c++17 -ffp-contract=off -fno-omit-frame-pointer"
g++ $SAN -fsanitize=address,undefined -I$R/raytracing_amd/csrc $R/scripts/sanitize/bvh_san.cpp $R/raytracing_amd/csrc/rt_bvh.cpp -o $T/bvh_asan -lpthread
g++ $SAN -fsanitize=thread -I$R/raytracing_amd/csrc $R/scripts/sanitize/bvh_san.cpp $R/raytracing_amd/csrc/rt_bvh.cpp -o $T/bvh_tsan -lpthread
g++ $SAN -fsanitize=address,undefined $R/scripts/sanitize/host_san.cpp $R/raytracing_amd/host/{raytracing,scenes,capi,image_io}.cpp \
    -L$R/raytracing_amd/csrc -lrt_amd -Wl,-rpath,$R/raytracing_amd/csrc -o $T/host_asan
g++ $SAN -fsanitize=address,undefined $R/scripts/sanitize/orc_san.cpp $R/oracle/oracle_capi.cpp -I$R/oracle -I$R/include -lpthread -o $T/orc_asan
for b in bvh_asan bvh_tsan host_asan orc_asan; do echo "== $b"; $T/$b; done
rm -rf $T
