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

// The form of a four-child node that the kernels read: 64 bytes, the children's boxes as 8-bit coordinates on a grid of
// 255 steps per axis that spans the node's own box (a little more than the union of its children).  A child's box on
// that grid always contains the child's box as the builder made it, with a margin that covers the rounding of the
// kernels' arithmetic on the grid coordinates: the node test decides which children to LOOK at, never what is hit.
//   lo[a] + q * step[a]   is the coordinate of grid line q (0..255) on axis a
//   qlo[a], qhi[a]        four bytes each: byte k is child k's lower / upper grid line on axis a
//   ref[k]                child k's reference (rt_device.h); an empty slot repeats child 0's reference under a box
//                         that is inside out (qlo = 255, qhi = 0), which no ray with a usable direction enters
// Half the bytes of the builder's own node per visit: what the extend kernel's tree walk is bound by is the rate at
// which a CU's vector memory path serves per-lane 16-byte reads at scattered addresses (about one per cycle).
struct PackedNode {
    float lo[3];
    float step_x;
    float step_y, step_z;
    uint32_t qlo_x, qhi_x;
    uint32_t qlo_y, qhi_y, qlo_z, qhi_z;
    uint32_t ref[4];
};
static_assert(sizeof(PackedNode) == 64, "four 16-byte groups");
#if RT_BVH_WIDTH == 4
std::vector<PackedNode> pack_nodes(const FlatBvh& bvh);
// Children of packed nodes whose grid box does not contain the builder's box with the margin pack_nodes promises (0 for
// a correct packing), and the mean ratio of a child's grid box surface to its own box surface (how much the grid costs).
struct PackCheck { uint32_t violations; double mean_surface_ratio; };
PackCheck check_packed_nodes(const FlatBvh& bvh, const std::vector<PackedNode>& packed);
#endif

}  // namespace rt
