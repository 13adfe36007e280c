#include "rt_bvh.h"
#include <cstdio>
#include <random>
#include <cstring>
#include <cmath>
using namespace rt;
static void run(const char* name, std::vector<BuildPrim> p, int threads) {
    FlatBvh b = build_bvh(p, threads);
    // walk: every leaf reachable once
    std::vector<int> seen(p.size(), 0);
    size_t found = 0;
    std::vector<uint32_t> todo;
    if (b.root != 0xFFFFFFFFu) todo.push_back(b.root);
    while (!todo.empty()) {
        uint32_t r = todo.back(); todo.pop_back();
        if (r & 0x80000000u) { found++; continue; }
        if (r >= b.n_nodes) { printf("%s: bad node ref %u\n", name, r); return; }
        for (int k = 0; k < BVH_WIDTH; ++k) {
            float slack = b.nodes[(size_t)NODE_FLOATS * r + NODE_SLACK_AT + k];
            if (!(slack > 0.0f)) continue;  // empty slot (NaN)
            uint32_t c; std::memcpy(&c, &b.nodes[(size_t)NODE_FLOATS * r + NODE_REFS_AT + k], 4);
            todo.push_back(c);
        }
    }
    printf("%s: n=%zu nodes=%u depth=%u stack=%u leaves found=%zu %s\n", name, p.size(), b.n_nodes, b.max_depth, b.stack_need, found, found == p.size() ? "ok" : "MISMATCH");
}
int main() {
    std::mt19937 g(1);
    std::uniform_real_distribution<float> U(0, 1);
    auto mk = [&](size_t n, int mode) {
        std::vector<BuildPrim> v(n);
        for (size_t i = 0; i < n; ++i) {
            float c[3] = {U(g) * 1000, U(g) * 1000, U(g) * 1000}, r = 0.05f + U(g);
            if (mode == 1) { c[0] = c[1] = c[2] = 5; r = 1; }              // all identical
            if (mode == 2) { c[1] = c[2] = 0; c[0] = (float)i * 1e-3f; r = 0; }  // degenerate points on a line
            if (mode == 3) { c[0] = std::ldexp(1.0f, (int)(i % 60)); c[1] = c[2] = 0; r = c[0] * 0.25f; }  // chain-like scales
            for (int a = 0; a < 3; ++a) { v[i].bmin[a] = c[a] - r; v[i].bmax[a] = c[a] + r; }
            v[i].ref = 0x80000000u | (uint32_t)i; v[i].exact = i & 1;
        }
        return v;
    };
    run("empty", {}, 0);
    run("one", mk(1, 0), 0);
    run("two", mk(2, 0), 0);
    run("five", mk(5, 0), 0);
    run("random 3000", mk(3000, 0), 0);
    run("identical 1000", mk(1000, 1), 0);
    run("line 2000", mk(2000, 2), 0);
    run("scales 600", mk(600, 3), 0);
    run("random 300k threaded", mk(300000, 0), 4);
    return 0;
}
