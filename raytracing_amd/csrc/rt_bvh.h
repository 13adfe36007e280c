// rt_bvh.h — host-side builder of the flattened BVH that the extend kernel walks.
// Replaces the pointer tree of Hittable::bvh (src/hittable.rs:103-155): closest
// hit does not depend on tree topology, so the split is binned SAH rather than
// the reference's median split.
#pragma once
#include <cstdint>
#include <vector>

namespace rt {

struct BuildPrim {
    float bmin[3], bmax[3];
    uint32_t ref;     // leaf reference stored in the tree
    uint32_t exact;   // 1: store the box as given (the reference's own leaf AABB); 0: widen it a little
};

// 16 floats per node, layout documented in rt_device.h.
struct FlatBvh {
    std::vector<float> nodes;  // 16 * n_nodes
    uint32_t root = 0xFFFFFFFFu;
    uint32_t n_nodes = 0;
    uint32_t max_depth = 0;
};

// Builds over `prims` (reordered in place).  threads <= 0 picks a default.
FlatBvh build_bvh(std::vector<BuildPrim>& prims, int threads);

}  // namespace rt
