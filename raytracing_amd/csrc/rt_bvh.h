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

// Children per node of the tree the kernels walk.  The builder always makes a binary tree first
// (binned SAH); for width 4 every other level is then folded into its parent.
#ifndef RT_BVH_WIDTH
#define RT_BVH_WIDTH 4
#endif
constexpr int BVH_WIDTH = RT_BVH_WIDTH;
static_assert(BVH_WIDTH == 2 || BVH_WIDTH == 4, "the BVH has 2 or 4 children per node");
constexpr int NODE_FLOATS = 8 * BVH_WIDTH;          // layout documented in rt_device.h
constexpr int NODE_REFS_AT = 6 * BVH_WIDTH;         // float index of the first child reference
constexpr int NODE_SLACK_AT = 7 * BVH_WIDTH;        // float index of the first child's slack factor (an empty slot has slack NaN)

struct FlatBvh {
    std::vector<float> nodes;  // NODE_FLOATS * n_nodes
    uint32_t root = 0xFFFFFFFFu;
    uint32_t n_nodes = 0;
    uint32_t max_depth = 0;
    // Entries a near-child-first traversal can have stacked at once in the worst case (every child of every node
    // on a root-to-leaf path is hit): max over such paths of the sum of (children - 1).
    uint32_t stack_need = 0;
};
// What the kernels' traversal stacks hold (LDS part + overflow area, or the per-lane array of the record entry points).
constexpr uint32_t TRAVERSAL_STACK_ENTRIES = 64;

// Builds over `prims` (reordered in place).  threads <= 0 picks a default.
FlatBvh build_bvh(std::vector<BuildPrim>& prims, int threads);

}  // namespace rt
