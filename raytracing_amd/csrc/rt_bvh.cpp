// rt_bvh.cpp — builder of the flattened BVH: a binary tree by binned SAH (one primitive per
// leaf; the n-1 inner nodes of an n-leaf tree are preallocated and handed out through an
// atomic counter so that subtrees can be built by several host threads), then folded to the
// four-children, 128-byte node layout of rt_device.h and renumbered.
#include "rt_bvh.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

namespace rt {
namespace {

constexpr int NBINS = 16;

struct Box {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* bmin, const float* bmax) {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], bmin[a]); mx[a] = std::max(mx[a], bmax[a]); }
    }
    void grow(const Box& b) { grow(b.mn, b.mx); }
    void grow_point(const float* p) { grow(p, p); }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

inline void centroid(const BuildPrim& p, float* c) {
    for (int a = 0; a < 3; ++a) c[a] = 0.5f * (p.bmin[a] + p.bmax[a]);
}

struct Builder {
    std::vector<BuildPrim>& prims;
    float* nodes;
    std::atomic<uint32_t> next_node{0};
    std::atomic<uint32_t> max_depth{0};
    uint32_t sah_depth = 40;  // below this depth ranges are split by count (a balanced subtree) instead of by SAH

    struct Task { size_t begin, end; uint32_t node; uint32_t depth; };

    explicit Builder(std::vector<BuildPrim>& p, float* n) : prims(p), nodes(n) {}

    Box bounds(size_t b, size_t e) const {
        Box bx;
        for (size_t i = b; i < e; ++i) bx.grow(prims[i].bmin, prims[i].bmax);
        return bx;
    }

    // Chooses the split of [b, e) and partitions; returns the middle index.
    size_t split(size_t b, size_t e, uint32_t depth) {
        Box cb;
        for (size_t i = b; i < e; ++i) { float c[3]; centroid(prims[i], c); cb.grow_point(c); }
        float best_cost = FLT_MAX;
        int best_axis = -1, best_bin = -1;
        // SAH follows the geometry and can peel off one primitive per level (outliers at growing distances); past
        // sah_depth ranges are split by count, which bounds the depth by sah_depth + log2(n).  Whether the stack of the
        // traversal suffices is checked on the finished tree (stack_need); build_bvh lowers sah_depth when it does not.
        for (int axis = 0; axis < 3 && depth < sah_depth; ++axis) {
            float lo = cb.mn[axis], hi = cb.mx[axis];
            if (!(hi > lo)) continue;
            float scale = (float)NBINS / (hi - lo);
            Box bb[NBINS];
            size_t cnt[NBINS] = {0};
            for (size_t i = b; i < e; ++i) {
                float c = 0.5f * (prims[i].bmin[axis] + prims[i].bmax[axis]);
                int k = std::min(NBINS - 1, std::max(0, (int)((c - lo) * scale)));
                bb[k].grow(prims[i].bmin, prims[i].bmax);
                cnt[k]++;
            }
            float right_area[NBINS];
            size_t right_cnt[NBINS];
            Box acc;
            size_t c = 0;
            for (int k = NBINS - 1; k > 0; --k) { acc.grow(bb[k]); c += cnt[k]; right_area[k] = acc.half_area(); right_cnt[k] = c; }
            Box left;
            size_t lc = 0;
            for (int k = 0; k < NBINS - 1; ++k) {
                left.grow(bb[k]); lc += cnt[k];
                if (lc == 0 || right_cnt[k + 1] == 0) continue;
                float cost = left.half_area() * (float)lc + right_area[k + 1] * (float)right_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = k; }
            }
        }
        size_t mid;
        if (best_axis < 0) {
            // coincident centroids or depth cap: median split along the widest centroid axis
            int axis = 0;
            for (int a = 1; a < 3; ++a) if (cb.mx[a] - cb.mn[a] > cb.mx[axis] - cb.mn[axis]) axis = a;
            mid = b + (e - b) / 2;
            std::nth_element(prims.begin() + (ptrdiff_t)b, prims.begin() + (ptrdiff_t)mid, prims.begin() + (ptrdiff_t)e,
                             [axis](const BuildPrim& x, const BuildPrim& y) { return x.bmin[axis] + x.bmax[axis] < y.bmin[axis] + y.bmax[axis]; });
        } else {
            float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
            float scale = (float)NBINS / (hi - lo);
            auto it = std::partition(prims.begin() + (ptrdiff_t)b, prims.begin() + (ptrdiff_t)e, [&](const BuildPrim& p) {
                float c = 0.5f * (p.bmin[best_axis] + p.bmax[best_axis]);
                int k = std::min(NBINS - 1, std::max(0, (int)((c - lo) * scale)));
                return k <= best_bin;
            });
            mid = (size_t)(it - prims.begin());
            if (mid == b || mid == e) mid = b + (e - b) / 2;
        }
        return mid;
    }

    // Boxes of inner nodes and of conservative leaves are widened so that rounding in the
    // slab test can never cull a box the exact ray touches; exact leaves keep the box the
    // reference itself would test (AABB::pad_to_minimums already applied by the caller).
    void write_child(uint32_t node, int side, Box bx, uint32_t ref, bool widen) {
        if (widen) {
            for (int a = 0; a < 3; ++a) {
                float m = std::max(std::fabs(bx.mn[a]), std::fabs(bx.mx[a]));
                float eps = std::max(5e-5f, m * 1e-6f);
                bx.mn[a] -= eps;
                bx.mx[a] += eps;
            }
        }
        // layout: see rt_device.h
        float* n = nodes + 16 * (size_t)node;
        float* xy = n + 4 * side;
        xy[0] = bx.mn[0]; xy[1] = bx.mn[1]; xy[2] = bx.mx[0]; xy[3] = bx.mx[1];
        n[8 + 2 * side] = bx.mn[2];
        n[9 + 2 * side] = bx.mx[2];
        std::memcpy(&n[12 + side], &ref, 4);
        n[14 + side] = widen ? 1.0000002f : 1.0f;
    }

    // Fills node `node` for range [b, e), e - b >= 2.  With `deferred` set, subranges of at
    // most `spawn_above` primitives are queued there (to be built by worker threads)
    // instead of being recursed into.
    void build_node(uint32_t node, size_t b, size_t e, uint32_t depth, size_t spawn_above, std::vector<Task>* deferred) {
        uint32_t d = max_depth.load(std::memory_order_relaxed);
        while (depth > d && !max_depth.compare_exchange_weak(d, depth)) {}
        size_t mid = split(b, e, depth);
        size_t rb[2] = {b, mid}, re[2] = {mid, e};
        for (int side = 0; side < 2; ++side) {
            size_t cnt = re[side] - rb[side];
            Box bx = bounds(rb[side], re[side]);
            if (cnt == 1) {
                write_child(node, side, bx, prims[rb[side]].ref, prims[rb[side]].exact == 0);
            } else {
                uint32_t child = next_node.fetch_add(1);
                write_child(node, side, bx, child, true);
                if (deferred && cnt <= spawn_above) deferred->push_back(Task{rb[side], re[side], child, depth + 1});
                else build_node(child, rb[side], re[side], depth + 1, spawn_above, deferred);
            }
        }
    }
};

}  // namespace

// Folds the binary tree into one with four children per node: a node takes over children of its inner children,
// whichever way gives the whole tree the least summed box surface of its nodes.  Boxes, references and slack factors
// are copied unchanged, so what is tested for a leaf does not depend on the width.
// Wide node (32 floats): child k's min.x min.y max.x max.y at 4k; z ranges (min, max) at 16 + 2k;
// references at 24 + k; slack at 28 + k (NaN marks an empty slot: no comparison with it holds, so such a child is never entered
// whatever the ray's interval - a zero would let a ray with t_min <= 0 into the slot's all-zero box).
static void fold_to_four(FlatBvh& bvh) {
    const uint32_t n2 = bvh.n_nodes;
    if (n2 == 0) return;
    const std::vector<float> in = std::move(bvh.nodes);
    struct Child { float xy[4], z[2], slack; uint32_t ref; };
    auto child_of = [&](uint32_t node, int side) {
        const float* n = &in[16 * (size_t)node];
        Child c;
        std::memcpy(c.xy, n + 4 * side, 16);
        c.z[0] = n[8 + 2 * side];
        c.z[1] = n[9 + 2 * side];
        std::memcpy(&c.ref, n + 12 + side, 4);
        c.slack = n[14 + side];
        return c;
    };
    std::vector<float> out;
    out.reserve(in.size());
    // Which binary nodes become wide nodes: chosen so that the summed box surface of the wide nodes - the expected number of
    // node visits of a random ray - is smallest.  T[n][j]: least cost of the subtree below binary node n when it may take up
    // to j + 1 slots of its parent; either n becomes a wide node itself (one slot; its own surface plus the cost of spreading its
    // subtree over four slots), or its two sides share the slots (split[n][j] of them go to the left side).
    std::vector<std::array<float, 4>> T(n2, std::array<float, 4>{0, 0, 0, 0});
    std::vector<std::array<uint8_t, 4>> split(n2, std::array<uint8_t, 4>{0, 0, 0, 0}), as_node(n2, std::array<uint8_t, 4>{1, 0, 0, 0});
    {
        auto surface = [&](uint32_t node) {
            const Child a = child_of(node, 0), b = child_of(node, 1);
            const double dx = (double)std::max(a.xy[2], b.xy[2]) - std::min(a.xy[0], b.xy[0]), dy = (double)std::max(a.xy[3], b.xy[3]) - std::min(a.xy[1], b.xy[1]),
                         dz = (double)std::max(a.z[1], b.z[1]) - std::min(a.z[0], b.z[0]);
            return dx * dy + dy * dz + dz * dx;
        };
        // children after parents in `post`, so walking it backwards sees children first
        std::vector<uint32_t> post{bvh.root};
        for (size_t i = 0; i < post.size(); ++i)
            for (int side = 0; side < 2; ++side) {
                const uint32_t r = child_of(post[i], side).ref;
                if (!(r & 0x80000000u)) post.push_back(r);
            }
        auto cost = [&](uint32_t ref, int slots) { return (ref & 0x80000000u) ? 0.0 : (double)T[ref][(size_t)slots - 1]; };
        for (size_t i = post.size(); i-- > 0;) {
            const uint32_t n = post[i], l = child_of(n, 0).ref, r = child_of(n, 1).ref;
            double forest[5] = {0, 0, 0, 0, 0};
            uint8_t best[5] = {0, 0, 1, 1, 1};
            for (int k = 2; k <= 4; ++k) {
                forest[k] = cost(l, 1) + cost(r, k - 1);
                for (int li = 2; li < k; ++li) {
                    const double c = cost(l, li) + cost(r, k - li);
                    if (c < forest[k]) { forest[k] = c; best[k] = (uint8_t)li; }
                }
            }
            const double own = surface(n) + forest[4];
            T[n][0] = (float)own;
            for (int k = 2; k <= 4; ++k) {
                const bool node = own <= forest[k];
                T[n][(size_t)k - 1] = (float)(node ? own : forest[k]);
                as_node[n][(size_t)k - 1] = node;
                split[n][(size_t)k - 1] = best[k];
            }
            split[n][0] = best[4];  // how a wide node at n deals out its four slots
        }
    }
    std::vector<uint32_t> new_of(n2, 0xFFFFFFFFu);
    // pass 1: which binary nodes survive (the root, and every inner node that ends up as a child slot), in visiting order
    std::vector<uint32_t> order{bvh.root};
    new_of[bvh.root] = 0;
    std::vector<std::array<Child, 4>> wide;
    std::vector<int> wide_n;
    uint32_t depth_max = 0;
    std::vector<uint32_t> depth{1};
    for (size_t head = 0; head < order.size(); ++head) {
        const uint32_t node = order[head];
        std::array<Child, 4> slots{};
        int k = 0;
        // deal the node's four slots out as the table says
        struct Item { uint32_t parent; int side, slots; };
        Item todo[8] = {{node, 1, 4 - split[node][0]}, {node, 0, split[node][0]}};  // at most four items wait at a time
        int n_todo = 2;
        while (n_todo > 0) {
            const Item it = todo[--n_todo];
            const Child c = child_of(it.parent, it.side);
            if ((c.ref & 0x80000000u) || it.slots == 1 || as_node[c.ref][(size_t)it.slots - 1]) { slots[k++] = c; continue; }
            const int li = split[c.ref][(size_t)it.slots - 1];
            todo[n_todo++] = Item{c.ref, 1, it.slots - li};
            todo[n_todo++] = Item{c.ref, 0, li};
        }
        for (int i = 0; i < k; ++i) {
            if (!(slots[i].ref & 0x80000000u)) {
                new_of[slots[i].ref] = (uint32_t)order.size();
                order.push_back(slots[i].ref);
                depth.push_back(depth[head] + 1);
            }
        }
        depth_max = std::max(depth_max, depth[head]);
        wide.push_back(slots);
        wide_n.push_back(k);
    }
    out.assign(32 * wide.size(), 0.0f);
    for (size_t i = 0; i < wide.size(); ++i) {
        float* n = &out[32 * i];
        for (int k = 0; k < 4; ++k) {
            Child c = wide[i][k];
            if (k >= wide_n[i]) {  // empty slot
                c = Child{};
                c.ref = 0xFFFFFFFFu;
                c.slack = std::numeric_limits<float>::quiet_NaN();
            } else if (!(c.ref & 0x80000000u)) {
                c.ref = new_of[c.ref];
            }
            std::memcpy(n + 4 * k, c.xy, 16);
            n[16 + 2 * k] = c.z[0];
            n[17 + 2 * k] = c.z[1];
            std::memcpy(n + 24 + k, &c.ref, 4);
            n[28 + k] = c.slack;
        }
    }
    bvh.nodes.swap(out);
    bvh.n_nodes = (uint32_t)wide.size();
    bvh.root = 0;
    bvh.max_depth = depth_max;
}

// Renumbers the nodes: breadth-first for the first BREADTH_FIRST_NODES (the top of the tree,
// which every ray walks, becomes one contiguous block at the start of the array), then each
// remaining subtree depth-first, parents before children.
static void renumber(FlatBvh& bvh) {
    constexpr uint32_t BREADTH_FIRST_NODES = 1024;
    const uint32_t n = bvh.n_nodes;
    if (n == 0) return;
    auto child = [&](uint32_t node, int k) {
        uint32_t r;
        std::memcpy(&r, &bvh.nodes[NODE_FLOATS * (size_t)node + NODE_REFS_AT + k], 4);
        return r;
    };
    std::vector<uint32_t> order;  // order[new] = old
    order.reserve(n);
    std::vector<uint32_t> frontier{bvh.root};
    size_t head = 0;
    while (head < frontier.size() && order.size() < BREADTH_FIRST_NODES) {
        uint32_t node = frontier[head++];
        order.push_back(node);
        for (int k = 0; k < BVH_WIDTH; ++k) {
            uint32_t r = child(node, k);
            if (!(r & 0x80000000u)) frontier.push_back(r);
        }
    }
    std::vector<uint32_t> stack;
    for (; head < frontier.size(); ++head) {
        stack.push_back(frontier[head]);
        while (!stack.empty()) {
            uint32_t node = stack.back();
            stack.pop_back();
            order.push_back(node);
            for (int k = BVH_WIDTH - 1; k >= 0; --k) {
                uint32_t r = child(node, k);
                if (!(r & 0x80000000u)) stack.push_back(r);
            }
        }
    }
    std::vector<uint32_t> new_of(n);
    for (uint32_t i = 0; i < (uint32_t)order.size(); ++i) new_of[order[i]] = i;
    std::vector<float> out(bvh.nodes.size());
    for (uint32_t i = 0; i < (uint32_t)order.size(); ++i) {
        const float* src = &bvh.nodes[NODE_FLOATS * (size_t)order[i]];
        float* dst = &out[NODE_FLOATS * (size_t)i];
        std::memcpy(dst, src, NODE_FLOATS * sizeof(float));
        for (int k = 0; k < BVH_WIDTH; ++k) {
            uint32_t r;
            std::memcpy(&r, &src[NODE_REFS_AT + k], 4);
            if (!(r & 0x80000000u)) {
                r = new_of[r];
                std::memcpy(&dst[NODE_REFS_AT + k], &r, 4);
            }
        }
    }
    bvh.nodes.swap(out);
    bvh.root = new_of[bvh.root];
    // worst-case stack use; children have larger numbers than their parents, so one backward pass does it
    std::vector<uint32_t> need(n, 0);
    for (uint32_t i = n; i-- > 0;) {
        uint32_t children = 0, deepest = 0;
        for (int k = 0; k < BVH_WIDTH; ++k) {
            if (!(bvh.nodes[NODE_FLOATS * (size_t)i + NODE_SLACK_AT + k] > 0.0f)) continue;  // empty slot (NaN)
            ++children;
            uint32_t r = child(i, k);
            if (!(r & 0x80000000u)) deepest = std::max(deepest, need[r]);
        }
        need[i] = (children ? children - 1 : 0) + deepest;
    }
    bvh.stack_need = need[bvh.root];
}

static FlatBvh build_with_sah_depth(std::vector<BuildPrim>& prims, int threads, uint32_t sah_depth) {
    FlatBvh out;
    const size_t n = prims.size();
    if (n == 0) return out;
    if (n == 1) {
        // no inner node at all: the root reference is the leaf itself.  (The reference's one-leaf
        // BVH node holds the leaf twice, the second time with an empty box, and an empty box
        // passes its min/max slab test, so the leaf is tested whatever the ray: same thing.)
        out.root = prims[0].ref;
        out.n_nodes = 0;
        out.max_depth = 0;
        return out;
    }
    out.nodes.assign(16 * (n - 1), 0.0f);
    Builder bld(prims, out.nodes.data());
    bld.sah_depth = sah_depth;
    uint32_t root = bld.next_node.fetch_add(1);
    if (threads <= 0) threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (threads == 1 || n < 200000) {
        bld.build_node(root, 0, n, 1, 0, nullptr);
    } else {
        std::vector<Builder::Task> tasks;
        bld.build_node(root, 0, n, 1, n / (size_t)(threads * 8), &tasks);
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= tasks.size()) break;
                bld.build_node(tasks[i].node, tasks[i].begin, tasks[i].end, tasks[i].depth, 0, nullptr);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
    }
    out.root = root;
    out.n_nodes = bld.next_node.load();
    out.max_depth = bld.max_depth.load();
    if (BVH_WIDTH == 4) fold_to_four(out);
    renumber(out);
    return out;
}

FlatBvh build_bvh(std::vector<BuildPrim>& prims, int threads) {
    // A chain of 40 one-versus-rest splits on top of a balanced tree of millions of leaves can ask for more stack entries
    // than the traversal has; less of the tree is then left to SAH (fewer chain levels), down to a plain split by count,
    // whose worst case is 3 entries per level of a tree of log4(n) levels.
    FlatBvh out;
    for (uint32_t sah_depth : {40u, 16u, 6u, 0u}) {
        out = build_with_sah_depth(prims, threads, sah_depth);
        if (out.stack_need <= TRAVERSAL_STACK_ENTRIES) break;
    }
    return out;
}

#if RT_BVH_WIDTH == 4
namespace {
// child k of the builder's wide node (layout: fold_to_four)
struct ChildBox { double mn[3], mx[3]; bool valid; uint32_t ref; };
ChildBox builder_child(const FlatBvh& bvh, uint32_t node, int k) {
    const float* n = &bvh.nodes[NODE_FLOATS * (size_t)node];
    ChildBox c;
    c.valid = n[NODE_SLACK_AT + k] > 0.0f;  // NaN: empty slot
    c.mn[0] = n[4 * k]; c.mn[1] = n[4 * k + 1]; c.mx[0] = n[4 * k + 2]; c.mx[1] = n[4 * k + 3];
    c.mn[2] = n[16 + 2 * k]; c.mx[2] = n[17 + 2 * k];
    std::memcpy(&c.ref, &n[NODE_REFS_AT + k], 4);
    return c;
}
// What a grid box must leave around the child's own box on an axis where the node spans `extent` and holds coordinates of
// magnitude up to `mag`: far more than the kernels' rounding (a few units in the last place of the extent and of the
// coordinates), far less than a grid step (extent / 255).
double pack_margin(double extent, double mag) { return std::ldexp(extent, -16) + std::ldexp(mag, -20) + 1e-30; }
}  // namespace

std::vector<PackedNode> pack_nodes(const FlatBvh& bvh) {
    std::vector<PackedNode> out(bvh.n_nodes);
    for (uint32_t i = 0; i < bvh.n_nodes; ++i) {
        ChildBox c[4];
        int first_valid = -1;
        for (int k = 0; k < 4; ++k) {
            c[k] = builder_child(bvh, i, k);
            if (c[k].valid && first_valid < 0) first_valid = k;
        }
        PackedNode& p = out[i];
        uint32_t* qlo[3] = {&p.qlo_x, &p.qlo_y, &p.qlo_z};
        uint32_t* qhi[3] = {&p.qhi_x, &p.qhi_y, &p.qhi_z};
        float* step[3] = {&p.step_x, &p.step_y, &p.step_z};
        for (int a = 0; a < 3; ++a) {
            double lo = DBL_MAX, hi = -DBL_MAX;
            for (int k = 0; k < 4; ++k)
                if (c[k].valid) { lo = std::min(lo, c[k].mn[a]); hi = std::max(hi, c[k].mx[a]); }
            if (first_valid < 0) lo = hi = 0.0;
            const double m = pack_margin(hi - lo, std::max(std::fabs(lo), std::fabs(hi)));
            const float lo_f = std::nextafterf((float)(lo - m), -INFINITY);
            const double top = hi + m;
            float s = std::nextafterf((float)((top - (double)lo_f) / 255.0), INFINITY);
            while ((double)lo_f + 255.0 * (double)s < top) s = std::nextafterf(s, INFINITY);
            p.lo[a] = lo_f;
            *step[a] = s;
            uint32_t lo_bytes = 0, hi_bytes = 0;
            for (int k = 0; k < 4; ++k) {
                int ql = 255, qh = 0;  // empty slot: inside out
                if (c[k].valid) {
                    ql = (int)std::floor((c[k].mn[a] - m - (double)lo_f) / (double)s);
                    ql = std::min(255, std::max(0, ql));
                    while (ql > 0 && (double)lo_f + ql * (double)s > c[k].mn[a] - m) --ql;
                    qh = (int)std::ceil((c[k].mx[a] + m - (double)lo_f) / (double)s);
                    qh = std::min(255, std::max(0, qh));
                    while (qh < 255 && (double)lo_f + qh * (double)s < c[k].mx[a] + m) ++qh;
                }
                lo_bytes |= (uint32_t)ql << (8 * k);
                hi_bytes |= (uint32_t)qh << (8 * k);
            }
            *qlo[a] = lo_bytes;
            *qhi[a] = hi_bytes;
        }
        for (int k = 0; k < 4; ++k) p.ref[k] = c[k].valid ? c[k].ref : (first_valid >= 0 ? c[first_valid].ref : 0xFFFFFFFFu);
    }
    return out;
}

PackCheck check_packed_nodes(const FlatBvh& bvh, const std::vector<PackedNode>& packed) {
    PackCheck r{0, 0.0};
    if (packed.size() != bvh.n_nodes) { r.violations = 1; return r; }
    double ratio_sum = 0.0;
    size_t ratio_n = 0;
    for (uint32_t i = 0; i < bvh.n_nodes; ++i) {
        const PackedNode& p = packed[i];
        const uint32_t qlo[3] = {p.qlo_x, p.qlo_y, p.qlo_z}, qhi[3] = {p.qhi_x, p.qhi_y, p.qhi_z};
        const double step[3] = {p.step_x, p.step_y, p.step_z};
        int first_valid = -1;
        for (int k = 0; k < 4; ++k) {
            const ChildBox c = builder_child(bvh, i, k);
            if (c.valid && first_valid < 0) first_valid = k;
            double own[3], grid[3];
            for (int a = 0; a < 3; ++a) {
                const int ql = (int)((qlo[a] >> (8 * k)) & 255u), qh = (int)((qhi[a] >> (8 * k)) & 255u);
                if (!c.valid) {
                    if (ql != 255 || qh != 0) ++r.violations;
                    continue;
                }
                const double g0 = (double)p.lo[a] + ql * step[a], g1 = (double)p.lo[a] + qh * step[a];
                // half of what pack_nodes leaves: the check's own arithmetic is exact, the kernels' is not
                const double half_m = 0.5 * std::ldexp(c.mx[a] - c.mn[a], -16);
                if (!(g0 <= c.mn[a] - half_m) || !(g1 >= c.mx[a] + half_m) || !(step[a] > 0.0) || !std::isfinite(step[a])) ++r.violations;
                own[a] = c.mx[a] - c.mn[a];
                grid[a] = g1 - g0;
            }
            if (c.valid) {
                if (p.ref[k] != c.ref) ++r.violations;
                const double so = own[0] * own[1] + own[1] * own[2] + own[2] * own[0], sg = grid[0] * grid[1] + grid[1] * grid[2] + grid[2] * grid[0];
                if (so > 0.0) { ratio_sum += sg / so; ++ratio_n; }
            }
        }
        for (int k = 0; k < 4; ++k)
            if (!builder_child(bvh, i, k).valid && first_valid >= 0 && p.ref[k] != builder_child(bvh, i, first_valid).ref) ++r.violations;
    }
    r.mean_surface_ratio = ratio_n ? ratio_sum / (double)ratio_n : 1.0;
    return r;
}
#endif

}  // namespace rt
