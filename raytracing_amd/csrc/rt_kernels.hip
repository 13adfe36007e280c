// rt_kernels.hip — gfx950 kernels of the wavefront path tracer.
//
// One sample of one pixel is a "path"; a path is advanced one segment per
// bounce by kernels that stream compacted queues through HBM:
//
//   raygen      get_ray (src/camera.rs:311-320,368-380)          -> ray queue
//   first leaves  the leaves kept out of the tree (scene-sized boxes), for every ray
//   extend      Ray::hit_anything (src/math/ray.rs:59-275): flattened-BVH closest
//               hit incl. constant media                          -> hit records
//   shade       ray_colour's loop body (src/camera.rs:339-364): Material::emitted /
//               scatter (src/material.rs:65-122), Texture::value (src/texture.rs:40-73),
//               PerlinNoise (src/perlin_noise.rs:39-102); finished paths write
//               their radiance, the rest are appended to the next ray queue
//   accumulate  acc[pixel] += radiance, in sample order (src/camera.rs:295-296)
//   finalize    the hdr / realtime per-pixel closures (src/camera.rs:95-100,115-118)
//
// Queue records are float4 / uint2 arrays indexed by queue slot, so every wave
// access is a contiguous 1 KiB (float4) or 512 B (uint2) segment.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_kernels.h"

// Leaf tests, box tests, ray generation and shading keep the reference's operation
// order (rustc never fuses a*b+c), so that hit distances and scattered rays agree
// with the reference's f32 results bit for bit wherever the same inputs are used.
#pragma clang fp contract(off)

namespace rt {

static constexpr float RT_INF = __builtin_huge_valf();
static constexpr int STACK_SIZE = (int)TRAVERSAL_STACK_ENTRIES;  // rt_scene_create rejects trees that could need more

// *p as one 16-byte load.  (Read as a struct, a record is four scalar loads to the compiler, which then moves the one for a
// component that is used later - a quad's D, a sphere's radius - down to that use: one more memory round trip in the middle
// of a leaf test.)
typedef float float4n __attribute__((ext_vector_type(4)));
RT_D float4 get4(const float4* p) {
    const float4n v = *(const float4n*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------------------
// leaf tests
// ---------------------------------------------------------------------------

// Both roots of Sphere::hit / MovingSphere::hit (src/hittable.rs:262-279), oc = origin - centre;
// false when the discriminant is negative.
RT_D bool sphere_roots(float3 centre, float radius, float3 o, float3 d, float a, float& near_root, float& far_root) {
    float3 oc = o - centre;
    float half_b = dot3(oc, d);
    float c = dot3(oc, oc) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    near_root = (-half_b - sq) / a;
    far_root = (-half_b + sq) / a;
    return true;
}

// Sphere::hit / MovingSphere::hit root selection (src/hittable.rs:262-279):
// oc = origin - centre, open interval tmin < root < tmax.
RT_D bool sphere_root(float3 centre, float radius, float3 o, float3 d, float a, float tmin, float tmax, float& t) {
    float3 oc = o - centre;
    float half_b = dot3(oc, d);
    float c = dot3(oc, oc) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    float root = (-half_b - sq) / a;
    if (!(tmin < root && root < tmax)) {
        root = (-half_b + sq) / a;
        if (!(tmin < root && root < tmax)) return false;
    }
    t = root;
    return true;
}

// The plane coordinates of Quad::hit (src/hittable.rs:389-391), in its operation order: alpha = w . (h x v),
// beta = w . (u x h) with h = hit point - q.  q4: the quad's record (layout in rt_device.h).
RT_D void quad_alpha_beta(float4 qq, float4 uu, float4 vv, float3 p, float& alpha, float& beta) {
    const float3 h = p - f3(qq), w = f3(qq.w, uu.w, vv.w);
    alpha = dot3(w, cross3(h, f3(vv)));
    beta = dot3(w, cross3(f3(uu), h));
}

// Quad::hit (src/hittable.rs:372-398) with the per-quad constants N, D, w hoisted:
// closed interval tmin <= t <= tmax, 0 <= alpha, beta <= 1.
// nd: the quad's first record (N, D), read by the caller - together with the leaf's placement where there is one.
// own_box (may be NULL): the quad's own AABB, min and max; a hit then only counts if own_box_test passes for it (see
// leaf_root).  inv = 1 / d, used by that test only.
RT_D bool own_box_test(float4 mn, float4 mx, float3 o, float3 inv, float lo, float hi);
RT_D bool own_box_test_of_walk(float4 mn, float4 mx, float3 o, float3 d, float3 inv, float lo, float hi);
template <bool ONE_TRIP = false>
RT_D bool quad_root(const float4* __restrict__ q4, float4 nd, float3 o, float3 d, float tmin, float tmax, float& t,
                    const float4* __restrict__ own_box = nullptr, float3 inv = make_float3(0.0f, 0.0f, 0.0f)) {
    // ONE_TRIP: the three remaining records (the same 64-byte line as the first) and the box are asked for before the plane
    // test has said that they are needed
    float4 qq, uu, vv, mn = make_float4(0, 0, 0, 0), mx = mn;
    if (ONE_TRIP) {
        qq = get4(q4 + 1), uu = get4(q4 + 2), vv = get4(q4 + 3);
        if (own_box) { mn = get4(own_box); mx = get4(own_box + 1); }
    }
    float3 N = f3(nd);
    float denom = dot3(N, d);
    if (fabsf(denom) < 1e-8f) return false;
    float tt = (nd.w - dot3(N, o)) / denom;
    if (!(tmin <= tt && tt <= tmax)) return false;
    if (!ONE_TRIP) {  // the three remaining records (and the quad's own box) in one round trip
        qq = get4(q4 + 1), uu = get4(q4 + 2), vv = get4(q4 + 3);
        if (own_box) { mn = get4(own_box); mx = get4(own_box + 1); }
    }
    const float3 h = (o + tt * d) - f3(qq), w = f3(qq.w, uu.w, vv.w);
    const float alpha = dot3(w, cross3(h, f3(vv)));
    if (!(0.0f <= alpha && alpha <= 1.0f)) return false;
    const float beta = dot3(w, cross3(f3(uu), h));
    if (!(0.0f <= beta && beta <= 1.0f)) return false;
    if (own_box && !own_box_test_of_walk(mn, mx, o, d, inv, tmin, tmax)) return false;
    t = tt;
    return true;
}

struct Probe {  // per-thread work counters (only live in the counting build)
    uint32_t nodes = 0, spheres = 0, quads = 0, media = 0;
    uint32_t max_sp = 0;  // deepest traversal stack this lane has had
};

// Move { RotateY { obj } } entry (src/math/ray.rs:207-229): origin - offset, then
// x' = cos*x - sin*z, z' = sin*x + cos*z for origin and direction.
RT_D void ray_to_object(float4 rot, float4 off, float3& o, float3& d) {
    float s = rot.x, c = rot.y;
    float3 o1 = o - f3(off);
    o = f3((c * o1.x) - (s * o1.z), o1.y, (s * o1.x) + (c * o1.z));
    d = f3((c * d.x) - (s * d.z), d.y, (s * d.x) + (c * d.z));
}
RT_D void ray_to_object(const DScene& sc, uint32_t xf, float3& o, float3& d) {
    ray_to_object(sc.xforms[2 * (size_t)xf], sc.xforms[2 * (size_t)xf + 1], o, d);
}
// The way back for a hit record (src/math/ray.rs:75-105): rotate point and normal,
// then add the offset to the point.
RT_D void hit_to_world(float4 rot, float4 off, float3& p, float3& n) {
    float s = rot.x, c = rot.y;
    p = f3((c * p.x) + (s * p.z), p.y, (-s * p.x) + (c * p.z)) + f3(off);
    n = f3((c * n.x) + (s * n.z), n.y, (-s * n.x) + (c * n.z));
}

// A leaf against a ray that is already in the leaf's own space (object space for a placed leaf).
template <bool COUNT>
RT_D bool leaf_root_in_leaf_space(const DScene& sc, uint32_t ref, float3 o, float3 d, float time, float a, float tmin, float tmax, float& t, Probe& pr) {
    const uint32_t kind = ref_kind(ref), idx = ref_index(ref);
    if (kind != KIND_QUAD) {
        if (COUNT) pr.spheres++;
        float4 s;
        float3 c;
        if (kind == KIND_SPHERE) {
            s = sc.spheres[idx];
            c = f3(s);
        } else {
            s = sc.moving[2 * (size_t)idx];
            c = f3(s) + time * f3(sc.moving[2 * (size_t)idx + 1]);  // MovingSphere::hit, src/hittable.rs:310
        }
        return sphere_root(c, s.w, o, d, a, tmin, tmax, t);
    } else {
        if (COUNT) pr.quads++;
        return quad_root(sc.quads + 4 * (size_t)idx, get4(sc.quads + 4 * (size_t)idx), o, d, tmin, tmax, t);
    }
}

// OWN_BOX: the leaf is a child of a tree node (not part of a medium's boundary).  With four-child nodes the node test is
// conservative, so the reference's test of the leaf's own box (AABBx2::hit, src/math/aabb.rs:127-167, on the box of
// Hittable::calc_aabb) is made here for every hit on a leaf that sits in world space: in the reference's tree that test
// decides whether the leaf is looked at at all, so a hit only counts where it passes.  It nearly always does (a ray that
// hits a primitive crosses its box); where roundings make the two disagree - a sphere smaller than the spacing of f32
// numbers at its distance, a quad's thin box far along a ray - the reference's outcome is the one reproduced.  A placed
// leaf's box is tested in object space by the reference, inside its Move / RotateY, and is not reproduced.  The box
// travels with the leaf's record (one round trip).  inv = 1 / d.
template <bool OWN_BOX> RT_D bool leaf_tests_own_box(uint32_t ref) {
#ifdef RT_NO_OWN_BOX  // timing experiments only (development builds): results differ where the reference's box test fails
    return false;
#else
    return OWN_BOX && RT_BVH_WIDTH == 4 && !ref_placed(ref);
#endif
}
// a sphere, still or moving: the two kinds share one copy of the root computation
template <bool COUNT, bool OWN_BOX = false>
RT_D bool sphere_leaf_root(const DScene& sc, uint32_t ref, float3 o, float3 d, float time, float a, float tmin, float tmax, float& t, Probe& pr,
                           float3 inv = make_float3(0.0f, 0.0f, 0.0f)) {
    const uint32_t kind = ref_kind(ref), idx = ref_index(ref);
    const bool own_box = leaf_tests_own_box<OWN_BOX>(ref);
    if (COUNT) pr.spheres++;
    float4 s;
    float3 c;
    const float4* xf;
    float4 mn = make_float4(0, 0, 0, 0), mx = mn;
    // (each kind reads its own arrays inside its branch: a pointer chosen by kind and used after the branches is a table
    // in scratch memory and a flat load to the compiler)
    // (a still sphere's box is not read where the scene says that it follows from the sphere's record: two gathers fewer)
    const bool computed_box = sc.plain_sphere_boxes != 0;
    if (kind == KIND_SPHERE) {
        s = get4(sc.spheres + idx);
        if (own_box && !computed_box) { mn = get4(sc.sphere_boxes + 2 * (size_t)idx); mx = get4(sc.sphere_boxes + 2 * (size_t)idx + 1); }
        c = f3(s);
        xf = sc.sphere_xf + 2 * (size_t)idx;
    } else {
        s = get4(sc.moving + 2 * (size_t)idx);
        const float4 dc = get4(sc.moving + 2 * (size_t)idx + 1);
        if (own_box) { mn = get4(sc.moving_boxes + 2 * (size_t)idx); mx = get4(sc.moving_boxes + 2 * (size_t)idx + 1); }
        c = f3(s) + time * f3(dc);  // MovingSphere::hit, src/hittable.rs:310
        xf = sc.moving_xf + 2 * (size_t)idx;
    }
    if (ref_placed(ref)) {
        ray_to_object(xf[0], xf[1], o, d);
        a = dot3(d, d);
    }
    if (!sphere_root(c, s.w, o, d, a, tmin, tmax, t)) return false;
    if (!own_box) return true;
    if (computed_box && kind == KIND_SPHERE) {  // AABB::from_points(centre - rvec, centre + rvec), src/hittable.rs:160-163
        const float ar = fabsf(s.w);
        mn = make_float4(s.x - ar, s.y - ar, s.z - ar, 0.0f);
        mx = make_float4(s.x + ar, s.y + ar, s.z + ar, 0.0f);
    }
    return own_box_test_of_walk(mn, mx, o, d, inv, tmin, tmax);
}
template <bool COUNT, bool OWN_BOX = false>
RT_D bool quad_leaf_root(const DScene& sc, uint32_t ref, float3 o, float3 d, float tmin, float tmax, float& t, Probe& pr,
                         float3 inv = make_float3(0.0f, 0.0f, 0.0f)) {
    const uint32_t idx = ref_index(ref);
    const bool own_box = leaf_tests_own_box<OWN_BOX>(ref);
    if (COUNT) pr.quads++;
    const float4* q4 = sc.quads + 4 * (size_t)idx;
    const float4 nd = get4(q4);  // on its way while the placement is read and applied
    if (ref_placed(ref)) ray_to_object(sc.quad_xf[2 * (size_t)idx], sc.quad_xf[2 * (size_t)idx + 1], o, d);
    // (in the tree walk the quads of a wave are tested together, a handful at a time: there one round trip for all of a quad's
    // records is worth more than the reads it saves to ask for three of them only after the plane test)
    return quad_root<OWN_BOX>(q4, nd, o, d, tmin, tmax, t, own_box ? sc.quad_boxes + 2 * (size_t)idx : nullptr, inv);
}
template <bool COUNT, bool OWN_BOX = false>
RT_D bool leaf_root(const DScene& sc, uint32_t ref, float3 o, float3 d, float time, float a, float tmin, float tmax, float& t, Probe& pr,
                    float3 inv = make_float3(0.0f, 0.0f, 0.0f)) {
    if (ref_kind(ref) != KIND_QUAD) return sphere_leaf_root<COUNT, OWN_BOX>(sc, ref, o, d, time, a, tmin, tmax, t, pr, inv);
    return quad_leaf_root<COUNT, OWN_BOX>(sc, ref, o, d, tmin, tmax, t, pr, inv);
}

// AABB::hit / AABBx2::hit (src/math/aabb.rs:32-49,127-167) for one box and the interval [lo, hi]: min / max that drop a
// NaN, the interval as the fourth lane of the reductions.  inv = 1 / direction, component by component.
RT_D bool own_box_test(float4 mn, float4 mx, float3 o, float3 inv, float lo, float hi) {
    const float x0 = (mn.x - o.x) * inv.x, x1 = (mx.x - o.x) * inv.x;
    const float y0 = (mn.y - o.y) * inv.y, y1 = (mx.y - o.y) * inv.y;
    const float z0 = (mn.z - o.z) * inv.z, z1 = (mx.z - o.z) * inv.z;
    const float t_in = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), fminf(lo, hi)));
    const float t_out = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), fmaxf(lo, hi)));
    return t_in <= t_out;
}

// The same with the tree walk's 1 / d (TravRay::inv), which differs from 1 / d only where it has been limited: the few rays
// that concerns take the divisions again.
RT_D bool own_box_test_of_walk(float4 mn, float4 mx, float3 o, float3 d, float3 inv, float lo, float hi) {
    if (fmaxf(fmaxf(fabsf(inv.x), fabsf(inv.y)), fabsf(inv.z)) >= 0x1p100f) inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    return own_box_test(mn, mx, o, inv, lo, hi);
}

// Closest hit of a medium's boundary object inside (lo, hi): one of the two nested queries of ConstantMediumPhase1/2
// (src/math/ray.rs:108-128,250-265).  A boundary of one leaf is that leaf.  A boundary of several leaves is a BVH in the
// reference (a box is the BVH of its six quads, src/hittable.rs:81-100), whose nodes test each leaf's own box - in the
// leaf's space, under its Move / RotateY - before the leaf.  For the thin box of a quad the outcome of that test matters:
// far along a ray, where t + 0.0001 == t, it is what keeps the second query from finding the first one's quad again.
// The leaves of a boundary nearly always share their placement (a placed box): the ray is then taken to their space
// once (ro, rd, ra: the ray there; rinv = 1 / rd), as the reference does on entering the Move.
template <bool COUNT>
RT_D bool boundary_root(const DScene& sc, const DMedium& m, float3 o, float3 d, float time, float a, bool shared, float3 ro, float3 rd,
                        float ra, float3 rinv, float lo, float hi, float& t, Probe& pr) {
    if (m.n_boundary == 1) return leaf_root<COUNT>(sc, sc.boundary_refs[m.first_boundary], o, d, time, a, lo, hi, t, pr);
    bool any = false;
    for (uint32_t i = 0; i < m.n_boundary; ++i) {
        const uint32_t ref = sc.boundary_refs[m.first_boundary + i];
        const float4* box = sc.boundary_boxes + 2 * (size_t)(m.first_boundary + i);
        float tt;
        bool hit;
        if (shared) {
            hit = own_box_test(box[0], box[1], ro, rinv, lo, hi) && leaf_root_in_leaf_space<COUNT>(sc, ref, ro, rd, time, ra, lo, hi, tt, pr);
        } else {
            float3 oo = o, dd = d;
            if (ref_placed(ref)) {
                const uint32_t kind = ref_kind(ref), idx = ref_index(ref);
                const float4* xf = (kind == KIND_SPHERE ? sc.sphere_xf : kind == KIND_MOVING ? sc.moving_xf : sc.quad_xf) + 2 * (size_t)idx;
                ray_to_object(xf[0], xf[1], oo, dd);
            }
            hit = own_box_test(box[0], box[1], oo, f3(1.0f / dd.x, 1.0f / dd.y, 1.0f / dd.z), lo, hi) &&
                  leaf_root<COUNT>(sc, ref, o, d, time, a, lo, hi, tt, pr);
        }
        if (hit) {
            any = true;
            hi = tt;
        }
    }
    t = hi;
    return any;
}

// A medium's record as three 16-byte loads that go out together (read field by field, the compiler fetches each field where
// it is first used: the density a memory round trip after the boundary sphere, in the middle of the longest test there is).
typedef uint32_t uint4n __attribute__((ext_vector_type(4)));
RT_D DMedium load_medium(const DMedium* p) {
    static_assert(sizeof(DMedium) == 48 && offsetof(DMedium, sphere) == 32, "three 16-byte groups");
    const uint4n a = *(const uint4n*)p, b = *((const uint4n*)p + 1);
    const float4n s = *((const float4n*)p + 2);
    DMedium m;
    m.first_boundary = a.x; m.n_boundary = a.y; m.neg_inv_density = __uint_as_float(a.z); m.phase_material = a.w;
    m.prim_id = b.x; m.single_sphere = b.y; m.shared_xform = b.z; m._pad = b.w;
    m.sphere = make_float4(s.x, s.y, s.z, s.w);
    return m;
}

// What a medium needs to start its random stream: (seed, pixel, sample) of the path and the
// bounce.  The render path only carries the path's index in the batch and derives the rest
// when a medium is actually entered.
struct MediumKey {
    uint64_t seed;
    uint32_t pixel, sample, bounce;
    RT_D uint64_t path_key_of() const { return path_key(seed, pixel, sample); }
};
struct BatchMediumKey {
    uint64_t seed;
    uint32_t path, bounce;  // path: the path's id in the batch (BatchInfo)
    BatchInfo b;
    RT_D uint64_t path_key_of() const {
        return path_key(seed, b.first_pixel + (path & ((1u << b.pixel_bits) - 1u)), b.first_sample + (path >> b.pixel_bits));
    }
};

// The boundary crossings of a medium whose boundary is one sphere, from the sphere's two roots: each is
// chosen exactly as Sphere::hit chooses it for the intervals (-inf, inf) and (0.0001 + t1, inf)
// (ConstantMediumPhase1, src/math/ray.rs:108-128).
RT_D bool medium_span_of_roots(float near_root, float far_root, float& t1, float& t2) {
    t1 = near_root;
    if (!(-RT_INF < t1 && t1 < RT_INF)) {
        t1 = far_root;
        if (!(-RT_INF < t1 && t1 < RT_INF)) return false;
    }
    float lo = 0.0001f + t1;
    t2 = near_root;
    if (!(lo < t2 && t2 < RT_INF)) {
        t2 = far_root;
        if (!(lo < t2 && t2 < RT_INF)) return false;
    }
    return true;
}

// ConstantMediumPhase2 (src/math/ray.rs:131-163): free-path sampling inside
// [t1, t2] clamped to the current interval.
template <class Key>
RT_D bool medium_sample(const DMedium& m, uint32_t idx, float t1, float t2, float a, float tmin, float tmax, const Key& key, float& t) {
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    t1 = fmaxf(t1, 0.0f);
    float ray_len = sqrtf(a);
    float distance_inside = (t2 - t1) * ray_len;
    Rng rng;
    rng.init(key.path_key_of(), key.bounce, STREAM_MEDIUM0 + idx);
    // f32::log(x, E as f32) = ln(x) / ln(2.7182817f) = ln(x) / 0.99999994f
    float hit_distance = m.neg_inv_density * (logf(rng.next()) / 0.99999994f);
    if (hit_distance > distance_inside) return false;
    t = t1 + hit_distance / ray_len;
    return true;
}

// GENERIC false: the scene's media are all bounded by one sphere in world space each (DScene::generic_media is 0), and the
// code for every other boundary stays out of the kernel - it costs the traversal kernel registers it does not have.
template <bool COUNT, bool GENERIC, class Key>
RT_D bool medium_root(const DScene& sc, uint32_t idx, float3 o, float3 d, float time, float a, float tmin, float tmax,
                      const Key& key, float& t, Probe& pr) {
    if (COUNT) pr.media++;
    DMedium m = load_medium(sc.media + idx);
    float t1, t2;
    if (!GENERIC || m.single_sphere) {
        // Boundary = one sphere in world space: the two queries share oc, half_b, c and the square root.
        if (COUNT) pr.spheres += 2;
        float near_root, far_root;
        if (!sphere_roots(f3(m.sphere), m.sphere.w, o, d, a, near_root, far_root)) return false;
        if (!medium_span_of_roots(near_root, far_root, t1, t2)) return false;
    } else {
        const bool shared = m.n_boundary > 1 && m.shared_xform != XFORM_MIXED;
        float3 ro = o, rd = d, rinv = o;
        float ra = a;
        if (shared) {
            if (m.shared_xform != REF_NONE) {  // REF_NONE = 0xFFFFFFFF: no placement
                ray_to_object(sc.xforms[2 * (size_t)m.shared_xform], sc.xforms[2 * (size_t)m.shared_xform + 1], ro, rd);
                ra = dot3(rd, rd);
            }
            rinv = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        }
        if (!boundary_root<COUNT>(sc, m, o, d, time, a, shared, ro, rd, ra, rinv, -RT_INF, RT_INF, t1, pr)) return false;
        if (!boundary_root<COUNT>(sc, m, o, d, time, a, shared, ro, rd, ra, rinv, 0.0001f + t1, RT_INF, t2, pr)) return false;
    }
    return medium_sample(m, idx, t1, t2, a, tmin, tmax, key, t);
}

// ---------------------------------------------------------------------------
// Traversal state and stacks.  A stack entry is (child reference, entry
// distance of its box); entries that start beyond the closest hit found so far
// are dropped when popped.
// ---------------------------------------------------------------------------
static constexpr uint32_t REF_DONE = 0xFFFFFFFEu;  // traversal finished (never a valid reference)

struct PrivateStack {  // per-lane array (scratch memory): used by the record-based entry points
    static constexpr bool ANY_INTERVAL = true;  // its rays come with the caller's interval: zero, negative, empty
    uint32_t ref[STACK_SIZE];
    float t[STACK_SIZE];
    int sp = 0;
    template <bool SHALLOW = false> RT_D void push(uint32_t r, float tt) { ref[sp] = r; t[sp] = tt; ++sp; }
    template <bool SHALLOW = false> RT_D bool pop(uint32_t& r, float& tt) {
        if (sp == 0) return false;
        --sp;
        r = ref[sp];
        tt = t[sp];
        return true;
    }
    RT_D bool wave_is_shallow(int) const { return false; }
    static constexpr bool HAS_LDS_POP = false;
    RT_D uint32_t pop_next_in_lds(float, uint32_t, uint32_t keep) { return keep; }
    RT_D uint32_t finish_node_in_lds(float, uint32_t, uint32_t r0, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) { return r0; }
};

// Stack of the render path's extend kernel.  An entry is 64 bits: (entry distance << 32) | ref.
// The first LEVELS levels live in LDS, laid out [level][lane] so that the 64 lanes of a
// wave always hit different banks whatever their stack depths; deeper levels (10 M-leaf trees)
// go to a per-wave area in global memory with the same layout.  The two address spaces are
// spelled out so that every access compiles to a plain ds_ or global_ instruction.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(1))) unsigned long long global_u64;
// OVERFLOW false: for trees whose walks never stack more than LEVELS entries (the builder's bound), no global part.
template <int LEVELS, bool OVERFLOW = true>
struct WaveStack {
    static constexpr bool ANY_INTERVAL = false;  // its rays are a render's: t_min = 0.001
    static constexpr bool HAS_LDS_POP = true;
    lds_u64* lds;     // this lane's column: level k at lds[64 * k]
    global_u64* ovf;  // the wave's area (the same pointer in all its lanes, so it lives in scalar registers): level LEVELS + k of lane l at ovf[64 * k + l]
    uint32_t lane;
    int sp = 0;
    // SHALLOW: the caller has checked (wave_is_shallow) that this access cannot reach the global part
    template <bool SHALLOW = false> RT_D void push(uint32_t r, float tt) {
        unsigned long long e = ((unsigned long long)__float_as_uint(tt) << 32) | r;
        if (!OVERFLOW || SHALLOW || sp < LEVELS) lds[64 * sp] = e;
        else ovf[64u * (uint32_t)(sp - LEVELS) + lane] = e;
        ++sp;
    }
    template <bool SHALLOW = false> RT_D bool pop(uint32_t& r, float& tt) {
        if (sp == 0) return false;
        --sp;
        unsigned long long e;
        if (!OVERFLOW || SHALLOW || sp < LEVELS) e = lds[64 * sp];
        else e = ovf[64u * (uint32_t)(sp - LEVELS) + lane];
        r = (uint32_t)e;
        tt = __uint_as_float((uint32_t)(e >> 32));
        return true;
    }
    // No lane of the wave is within `room` entries of the end of the LDS levels: up to `room` pushes, and any number of pops,
    // stay in LDS for every lane.  One comparison per node for the wave, instead of one branch per push and pop and lane:
    // the scalar instructions of those branches (exec-mask bookkeeping) are what the extend kernel's time is most sensitive to.
    RT_D bool wave_is_shallow(int room) const { return !OVERFLOW || __ballot(sp > LEVELS - room) == 0ull; }
    // pop_next (below) for the lanes with `need` set, entries in LDS only (wave_is_shallow(0)); the other lanes get `keep`.
    // Written out: the lanes that still look for an entry are the exec mask, a lane leaves when its stack is empty or its
    // entry can hold a closer hit, and the wave leaves when no lane is left - three scalar instructions and two branches a
    // round, where the compiler's code for the same loop keeps its own masks beside exec (fourteen).
    // The end of a node step in one piece, entries in LDS only (wave_is_shallow(3)): keys in ascending order with ~0 for a
    // child that was missed; the hit children other than the nearest go on the stack farthest first, a lane with no hit at
    // all pops (as pop_next_in_lds); returns the next reference.
    RT_D uint32_t finish_node_in_lds(float tmax, uint32_t k0, uint32_t r0, uint32_t k1, uint32_t r1, uint32_t k2, uint32_t r2, uint32_t k3,
                                     uint32_t r3) {
        uint32_t out = r0;
        uint32_t addr, lo, hi;
        unsigned long long saved, m1, m2, m3;
        const uint32_t base = (uint32_t)(uintptr_t)lds;
        asm volatile(
            // (every comparison while all of the step's lanes are active: a comparison leaves the bits of inactive lanes 0)
            "v_cmp_ne_u32_e64 %[m3], -1, %[k3]\n\t"
            "v_cmp_ne_u32_e64 %[m2], -1, %[k2]\n\t"
            "v_cmp_ne_u32_e64 %[m1], -1, %[k1]\n\t"
            "v_cmp_eq_u32 vcc, -1, %[k0]\n\t"
            "s_mov_b64 %[saved], exec\n\t"
            "s_and_b64 exec, %[saved], %[m3]\n\t"
            "v_lshl_add_u32 %[addr], %[sp], 9, %[base]\n\t"
            "ds_write2_b32 %[addr], %[r3], %[k3] offset1:1\n\t"
            "v_add_u32 %[sp], 1, %[sp]\n\t"
            "s_and_b64 exec, %[saved], %[m2]\n\t"
            "v_lshl_add_u32 %[addr], %[sp], 9, %[base]\n\t"
            "ds_write2_b32 %[addr], %[r2], %[k2] offset1:1\n\t"
            "v_add_u32 %[sp], 1, %[sp]\n\t"
            "s_and_b64 exec, %[saved], %[m1]\n\t"
            "v_lshl_add_u32 %[addr], %[sp], 9, %[base]\n\t"
            "ds_write2_b32 %[addr], %[r1], %[k1] offset1:1\n\t"
            "v_add_u32 %[sp], 1, %[sp]\n\t"
            "s_and_b64 exec, %[saved], vcc\n\t"
            "s_cbranch_execz 2f\n\t"
            "v_mov_b32 %[out], -2\n"
            "1:\n\t"
            "v_cmp_lt_i32 vcc, 0, %[sp]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execz 2f\n\t"
            "v_add_u32 %[sp], -1, %[sp]\n\t"
            "v_lshl_add_u32 %[addr], %[sp], 9, %[base]\n\t"
            "ds_read_b32 %[lo], %[addr]\n\t"
            "ds_read_b32 %[hi], %[addr] offset:4\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_le_f32 vcc, %[hi], %[tmax]\n\t"
            "v_cndmask_b32 %[out], %[out], %[lo], vcc\n\t"
            "s_andn2_b64 exec, exec, vcc\n\t"
            "s_cbranch_execnz 1b\n"
            "2:\n\t"
            "s_mov_b64 exec, %[saved]"
            : [out] "+v"(out), [sp] "+v"(sp), [addr] "=&v"(addr), [lo] "=&v"(lo), [hi] "=&v"(hi), [saved] "=&s"(saved), [m1] "=&s"(m1),
              [m2] "=&s"(m2), [m3] "=&s"(m3)
            : [k0] "v"(k0), [k1] "v"(k1), [r1] "v"(r1), [k2] "v"(k2), [r2] "v"(r2), [k3] "v"(k3), [r3] "v"(r3), [base] "v"(base), [tmax] "v"(tmax)
            : "vcc", "scc", "memory");
        static_assert(REF_DONE == 0xFFFFFFFEu, "the assembly above spells REF_DONE as -2");
        return out;
    }
    // The lanes that pop are those whose `probe` is ~0 (a node step hands in its nearest child's key: ~0 says no child was hit).
    RT_D uint32_t pop_next_in_lds(float tmax, uint32_t probe, uint32_t keep) {
        uint32_t out = probe == 0xFFFFFFFFu ? REF_DONE : keep;
        uint32_t addr, lo, hi;
        unsigned long long saved;
        const uint32_t base = (uint32_t)(uintptr_t)lds;
        asm volatile(
            "v_cmp_eq_u32 vcc, -1, %[probe]\n\t"
            "s_and_saveexec_b64 %[saved], vcc\n\t"
            "s_cbranch_execz 2f\n"
            "1:\n\t"
            "v_cmp_lt_i32 vcc, 0, %[sp]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execz 2f\n\t"
            "v_add_u32 %[sp], -1, %[sp]\n\t"
            "v_lshl_add_u32 %[addr], %[sp], 9, %[base]\n\t"
            "ds_read_b32 %[lo], %[addr]\n\t"
            "ds_read_b32 %[hi], %[addr] offset:4\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_le_f32 vcc, %[hi], %[tmax]\n\t"
            "v_cndmask_b32 %[out], %[out], %[lo], vcc\n\t"
            "s_andn2_b64 exec, exec, vcc\n\t"
            "s_cbranch_execnz 1b\n"
            "2:\n\t"
            "s_mov_b64 exec, %[saved]"
            : [out] "+v"(out), [sp] "+v"(sp), [addr] "=&v"(addr), [lo] "=&v"(lo), [hi] "=&v"(hi), [saved] "=&s"(saved)
            : [probe] "v"(probe), [base] "v"(base), [tmax] "v"(tmax)
            : "vcc", "scc", "memory");
        return out;
    }
};

// Start of the interval of every ray of a render (src/camera.rs:346: Interval::new(0.001, f32::MAX)); rays that come with
// the caller's own interval go through PrivateStack (ANY_INTERVAL).
static constexpr float RENDER_TMIN = 0.001f;
static_assert(__builtin_bit_cast(uint32_t, RENDER_TMIN) == 0x3a83126fu, "node_step spells this constant out in an instruction");
struct TravRay {
    float3 o, d;
    float3 inv;  // 1 / d, limited to +-INV_LIMIT
    float time, a, tmin;
};

// The walk's 1 / direction stays finite.  With an infinite one (a direction component that is zero: the middle row of an
// image whose camera looks along an axis plane has such rays) the grid form of the slab distances, q * (step / d) +
// (lo - o) / d, is inf - inf or 0 * inf for every child: the axis would drop out of the test, and the ray would visit
// every box along its line whatever their position on that axis.  With the limit, the distances of that axis are huge and of
// the right sign: beyond any tmax where the origin is outside the slab, on either side of the ray's interval where it is
// inside.  A component smaller than 2^-100 is treated as if it were larger: that only lets more children pass.
static constexpr float INV_LIMIT = 0x1p100f;
RT_D TravRay make_trav_ray(float3 o, float3 d, float time, float tmin) {
    TravRay r;
    r.o = o; r.d = d; r.time = time; r.tmin = tmin;
    r.a = dot3(d, d);
    r.inv = f3(__builtin_amdgcn_fmed3f(1.0f / d.x, -INV_LIMIT, INV_LIMIT), __builtin_amdgcn_fmed3f(1.0f / d.y, -INV_LIMIT, INV_LIMIT),
               __builtin_amdgcn_fmed3f(1.0f / d.z, -INV_LIMIT, INV_LIMIT));
    return r;
}

// Next reference to process after the current one is finished: pops until an entry that can
// still hold a closer hit, REF_DONE when the stack runs out.
template <bool SHALLOW = false, class Stack>
RT_D uint32_t pop_next(Stack& st, float tmax) {
    uint32_t r;
    float tt;
    while (st.template pop<SHALLOW>(r, tt))
        if (tt <= tmax) return r;
    return REF_DONE;
}

// Where inner nodes are read from: the array in global memory, or for the first n_top nodes
// (the top levels of the tree, numbered breadth-first) a copy in LDS.
typedef float float4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float4v lds_float4v;
RT_D float4 f4(float4v v) { return make_float4(v.x, v.y, v.z, v.w); }
struct NodeSource {
    const float4* nodes;
    const float4* top;   // LDS copy of nodes[0 .. n_top) (generic pointer to shared memory), or unused
    uint32_t n_top;
};
typedef float float2v __attribute__((ext_vector_type(2)));
static constexpr uint32_t NODE_FLOAT4S = 4;  // 64 bytes per node, whichever the width (rt_device.h)

// One inner node: slab-tests both children, continues with the nearer one and stacks the
// other.  Returns the next reference (inner node, leaf, or REF_DONE).
//
// The slab distances are (bound - origin) * (1 / direction), each rounded once, exactly the
// arithmetic of AABBx2::hit (src/math/aabb.rs:127-167); they are evaluated two at a time with
// packed subtract / multiply, which round like the scalar instructions.  A conservative box
// (inner node, placed leaf, medium) has its exit distance widened by the node's slack factor;
// the box of a leaf that sits in world space is the reference's own AABB and is tested with
// slack 1, so the same leaves get tested as in the reference.
#if RT_BVH_WIDTH == 2
template <bool COUNT, bool LDS_TOP, class Stack>
RT_D uint32_t node_step(const NodeSource& src, const TravRay& r, uint32_t cur, float tmax, Stack& st, Probe& pr) {
    if (COUNT) pr.nodes++;
    float4 n0, n1, n2, n3;
    if (LDS_TOP && cur < src.n_top) {
        const lds_float4v* n = (const lds_float4v*)src.top + 4 * cur;
        n0 = f4(n[0]); n1 = f4(n[1]); n2 = f4(n[2]); n3 = f4(n[3]);
    } else {
        const float4* n = src.nodes + 4 * (size_t)cur;
        n0 = n[0]; n1 = n[1]; n2 = n[2]; n3 = n[3];
    }
    float l_in, r_in, l_out, r_out;
    {
        const float2v oxy = {r.o.x, r.o.y}, ixy = {r.inv.x, r.inv.y}, ozz = {r.o.z, r.o.z}, izz = {r.inv.z, r.inv.z};
        // NaNs (0 * inf) drop out of fminf/fmaxf, as they do out of simd_min/simd_max in the reference
        const float2v l0 = (float2v{n0.x, n0.y} - oxy) * ixy, l1 = (float2v{n0.z, n0.w} - oxy) * ixy;
        const float2v r0 = (float2v{n1.x, n1.y} - oxy) * ixy, r1 = (float2v{n1.z, n1.w} - oxy) * ixy;
        const float2v lz = (float2v{n2.x, n2.y} - ozz) * izz, rz = (float2v{n2.z, n2.w} - ozz) * izz;
        l_in = fmaxf(fmaxf(fminf(l0.x, l1.x), fminf(l0.y, l1.y)), fmaxf(fminf(lz.x, lz.y), r.tmin));
        l_out = fminf(fminf(fmaxf(l0.x, l1.x), fmaxf(l0.y, l1.y)), fminf(fmaxf(lz.x, lz.y), tmax));
        r_in = fmaxf(fmaxf(fminf(r0.x, r1.x), fminf(r0.y, r1.y)), fmaxf(fminf(rz.x, rz.y), r.tmin));
        r_out = fminf(fminf(fmaxf(r0.x, r1.x), fmaxf(r0.y, r1.y)), fminf(fmaxf(rz.x, rz.y), tmax));
    }
    const uint32_t left = __float_as_uint(n3.x), right = __float_as_uint(n3.y);
    const bool hl = l_in <= l_out * n3.z;
    const bool hr = r_in <= r_out * n3.w;
    if (hl && hr) {
        bool left_first = l_in <= r_in;
        st.push(left_first ? right : left, left_first ? r_in : l_in);
        return left_first ? left : right;
    }
    if (hl) return left;
    if (hr) return right;
    return pop_next(st, tmax);
}
#else
// Four children per node, 64 bytes per node (rt_bvh.h, PackedNode): the children's boxes are 8-bit coordinates on a grid
// that spans the node's own box, q in 0..255 on axis a standing for lo[a] + q * step[a].  A child's slab distances come
// straight from the grid coordinates: t = q * (step / d) + (lo - o) / d, one fused multiply-add per bound after two
// multiplications and a subtraction per axis and node.  The words that hold the four children's lower / upper coordinates
// of an axis are swapped where the direction is negative, so the first is what the ray meets first on that axis and no
// min / max is spent on ordering the pair.
//
// This test is conservative, never exact: a child's grid box contains the child's box with a margin of 2^-16 of the
// node's extent (pack_nodes), against roundings here of a few 2^-24 of that extent, and the entry distance is shrunk by
// 2^-18 of itself, against roundings of a few 2^-24 of the distances.  It decides which children are looked at.  What
// is hit is decided by the leaves' own tests, and for a leaf that sits in world space by the reference's test of the
// leaf's own box on top (own_box_test_of_walk), as in the reference's tree.  1 / d is finite here (make_trav_ray), so a
// NaN distance takes a NaN or infinite origin, or an infinite coordinate; it drops out of the max / min reductions, as it
// does out of simd_min / simd_max in the reference, and so never culls a child.
// The children that are hit are ordered by entry distance with a five-exchange sorting network (entry distances
// are positive, so their bit patterns order like the numbers; a child that is missed gets the
// key ~0), the nearest is continued with and the others are stacked farthest first.
RT_D void order2(uint32_t& ka, uint32_t& ra, uint32_t& kb, uint32_t& rb) {
    const bool swap = kb < ka;
    const uint32_t k0 = swap ? kb : ka, k1 = swap ? ka : kb, r0 = swap ? rb : ra, r1 = swap ? ra : rb;
    ka = k0; kb = k1; ra = r0; rb = r1;
}
typedef __attribute__((address_space(3))) const char lds_bytes;
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const uint4v lds_uint4v;
// a * b[SEL_B] + c[SEL_C] for both halves of a, fused: a packed multiply-add that takes one half of its second and third
// operand for both lanes (operand selection is part of the instruction; written out, the compiler would copy the halves
// into register pairs of their own, and the extend kernel has no registers to spare).
template <int SEL_B, int SEL_C>
RT_D float2v pk_fma_half(float2v a, float2v b, float2v c) {
    float2v d;
    if (SEL_B == 0 && SEL_C == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else if (SEL_B == 1 && SEL_C == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
static constexpr float NODE_TEST_SHRINK = 1.0f - 0x1p-18f;   // what the entry distance of a child's grid box is multiplied by
template <bool COUNT, bool LDS_TOP, class Stack>
RT_D uint32_t node_step(const NodeSource& src, const TravRay& r, uint32_t cur, float tmax, Stack& st, Probe& pr) {
    if (COUNT) pr.nodes++;
    const uint32_t base = cur * 64u;
    uint4v g0, g1, g2, g3;
#ifdef RT_EXTRA_NODE_LOADS
    uint32_t extra_word = 0;
#endif
    if (LDS_TOP && cur < src.n_top) {
        lds_bytes* n = (lds_bytes*)src.top;
        g0 = *(const lds_uint4v*)(n + base); g1 = *(const lds_uint4v*)(n + (base + 16u));
        g2 = *(const lds_uint4v*)(n + (base + 32u)); g3 = *(const lds_uint4v*)(n + (base + 48u));
    } else {
        const char* n = (const char*)src.nodes;
#ifdef RT_EXTRA_NODE_LOADS  // timing experiments (development builds): more bytes through the L1 from the node's own cache line
#pragma unroll
        for (int x = 0; x < RT_EXTRA_NODE_LOADS; ++x) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(extra_word) : "v"(n + base), "n"(4 * x) : "memory");
#endif
        g0 = *(const uint4v*)(n + base); g1 = *(const uint4v*)(n + (base + 16u));
        g2 = *(const uint4v*)(n + (base + 32u)); g3 = *(const uint4v*)(n + (base + 48u));
    }
    // per axis: A = (lo - o) / d, S = step / d; x and y travel as pairs, z as (S, A)
    const float2v Axy = {(__uint_as_float(g0.x) - r.o.x) * r.inv.x, (__uint_as_float(g0.y) - r.o.y) * r.inv.y};
    const float2v Sxy = {__uint_as_float(g0.w) * r.inv.x, __uint_as_float(g1.x) * r.inv.y};
    const float2v SAz = {__uint_as_float(g1.y) * r.inv.z, (__uint_as_float(g0.z) - r.o.z) * r.inv.z};
    // the words of the sides met first / last: (lower, upper) as stored where the direction is positive, swapped where not
    const uint32_t sx = (uint32_t)((int32_t)__float_as_uint(r.inv.x) >> 31), sy = (uint32_t)((int32_t)__float_as_uint(r.inv.y) >> 31),
                   sz = (uint32_t)((int32_t)__float_as_uint(r.inv.z) >> 31);
    // entry (tn) and exit (tf) distance of child k along each axis
    float tn[3][4], tf[3][4];
#define RT_NODE_AXIS(AXIS, SIGN, LOWER, UPPER, SB, SC, BV, CV)                                                                        \
    {                                                                                                                                 \
        const uint32_t nw = ((UPPER) & (SIGN)) | ((LOWER) & ~(SIGN)), fw = ((LOWER) & (SIGN)) | ((UPPER) & ~(SIGN)); /* v_bfi_b32 */   \
        const float2v n01 = {(float)(nw & 255u), (float)((nw >> 8) & 255u)}, n23 = {(float)((nw >> 16) & 255u), (float)(nw >> 24)};   \
        const float2v f01 = {(float)(fw & 255u), (float)((fw >> 8) & 255u)}, f23 = {(float)((fw >> 16) & 255u), (float)(fw >> 24)};   \
        const float2v a0 = pk_fma_half<SB, SC>(n01, BV, CV), a1 = pk_fma_half<SB, SC>(n23, BV, CV);                                   \
        const float2v b0 = pk_fma_half<SB, SC>(f01, BV, CV), b1 = pk_fma_half<SB, SC>(f23, BV, CV);                                   \
        tn[AXIS][0] = a0.x; tn[AXIS][1] = a0.y; tn[AXIS][2] = a1.x; tn[AXIS][3] = a1.y;                                               \
        tf[AXIS][0] = b0.x; tf[AXIS][1] = b0.y; tf[AXIS][2] = b1.x; tf[AXIS][3] = b1.y;                                               \
    }
    RT_NODE_AXIS(0, sx, g1.z, g1.w, 0, 0, Sxy, Axy)
    RT_NODE_AXIS(1, sy, g2.x, g2.y, 1, 1, Sxy, Axy)
    RT_NODE_AXIS(2, sz, g2.z, g2.w, 0, 1, SAz, SAz)
#undef RT_NODE_AXIS
    // max / min over the axes and the ray's interval, NaNs dropping out: the instructions fmaxf / fminf compile to,
    // spelled out (in one piece: the compiler pads every separate piece of assembly with idle cycles against hazards it
    // cannot rule out) so that none is spent on quieting operands that are results of arithmetic
    float t_in[4], t_out[4];
    if (Stack::ANY_INTERVAL) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a, b;
            asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(tn[2][k]), "v"(r.tmin));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t_in[k]) : "v"(tn[0][k]), "v"(tn[1][k]), "v"(a));
            asm("v_min_f32 %0, %1, %2" : "=v"(b) : "v"(tf[2][k]), "v"(tmax));
            asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t_out[k]) : "v"(tf[0][k]), "v"(tf[1][k]), "v"(b));
        }
    } else {
        // RENDER_TMIN as a literal: no register holds it
        asm("v_max_f32 %0, 0x3a83126f, %8\n\tv_max_f32 %1, 0x3a83126f, %9\n\tv_max_f32 %2, 0x3a83126f, %10\n\tv_max_f32 %3, 0x3a83126f, %11\n\t"
            "v_min_f32 %4, %12, %16\n\tv_min_f32 %5, %13, %16\n\tv_min_f32 %6, %14, %16\n\tv_min_f32 %7, %15, %16"
            : "=&v"(t_in[0]), "=&v"(t_in[1]), "=&v"(t_in[2]), "=&v"(t_in[3]), "=&v"(t_out[0]), "=&v"(t_out[1]), "=&v"(t_out[2]), "=&v"(t_out[3])
            : "v"(tn[2][0]), "v"(tn[2][1]), "v"(tn[2][2]), "v"(tn[2][3]), "v"(tf[2][0]), "v"(tf[2][1]), "v"(tf[2][2]), "v"(tf[2][3]), "v"(tmax));
        asm("v_max3_f32 %0, %8, %12, %0\n\tv_max3_f32 %1, %9, %13, %1\n\tv_max3_f32 %2, %10, %14, %2\n\tv_max3_f32 %3, %11, %15, %3\n\t"
            "v_min3_f32 %4, %16, %20, %4\n\tv_min3_f32 %5, %17, %21, %5\n\tv_min3_f32 %6, %18, %22, %6\n\tv_min3_f32 %7, %19, %23, %7"
            : "+v"(t_in[0]), "+v"(t_in[1]), "+v"(t_in[2]), "+v"(t_in[3]), "+v"(t_out[0]), "+v"(t_out[1]), "+v"(t_out[2]), "+v"(t_out[3])
            : "v"(tn[0][0]), "v"(tn[0][1]), "v"(tn[0][2]), "v"(tn[0][3]), "v"(tn[1][0]), "v"(tn[1][1]), "v"(tn[1][2]), "v"(tn[1][3]),
              "v"(tf[0][0]), "v"(tf[0][1]), "v"(tf[0][2]), "v"(tf[0][3]), "v"(tf[1][0]), "v"(tf[1][1]), "v"(tf[1][2]), "v"(tf[1][3]));
    }
#if defined(RT_EXTRA_NODE_VALU) || defined(RT_EXTRA_NODE_SALU)  // timing experiments (development builds): instructions that do nothing
    {
#ifdef RT_EXTRA_NODE_VALU
        float sink = t_in[0];
#pragma unroll
        for (int x = 0; x < RT_EXTRA_NODE_VALU; ++x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sink) : "v"(t_out[x & 3]));
        asm volatile("" :: "v"(sink));
#endif
#ifdef RT_EXTRA_NODE_SALU
#pragma unroll
        for (int x = 0; x < RT_EXTRA_NODE_SALU; ++x) asm volatile("s_mov_b32 s50, 0" ::: "s50");
#endif
    }
#endif
    const uint32_t refs[4] = {g3.x, g3.y, g3.z, g3.w};
#ifdef RT_EXTRA_NODE_LOADS
    asm volatile("" :: "v"(extra_word), "v"(g3.x));  // (loads return in order: the extra ones have arrived once the node's last group has)
#endif
    uint32_t key[4], ref[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (Stack::ANY_INTERVAL) {
            // a caller's interval may reach below zero: a negative distance moves towards zero by 2^-18 of its own size where
            // a positive entry distance is shrunk (for positive distances: the decisions and keys of the render path)
            const float in = t_in[k] * (t_in[k] < 0.0f ? 2.0f - NODE_TEST_SHRINK : NODE_TEST_SHRINK);
            const float out = t_out[k] < 0.0f ? t_out[k] * NODE_TEST_SHRINK : t_out[k];
            key[k] = in <= out ? __float_as_uint(in) : 0xFFFFFFFFu;
        } else {
            // The entry distance shrunk by 2^-18 of itself is the key (a render's rays start at RENDER_TMIN: it is positive);
            // the sign of exit - entry, spread over the word, turns it into the key of a missed child.  inf - inf (a child
            // whose entry distance is infinite holds no hit either way) may fall on either side.
            const float in = t_in[k] * NODE_TEST_SHRINK;
            key[k] = __float_as_uint(in) | (uint32_t)((int32_t)__float_as_uint(t_out[k] - in) >> 31);
        }
        ref[k] = refs[k];
    }
    order2(key[0], ref[0], key[1], ref[1]);
    order2(key[2], ref[2], key[3], ref[3]);
    order2(key[0], ref[0], key[2], ref[2]);
    order2(key[1], ref[1], key[3], ref[3]);
    order2(key[1], ref[1], key[2], ref[2]);
    if (st.wave_is_shallow(3)) {  // the usual case: neither the pushes nor the pops of this node can leave the LDS levels
        if (Stack::HAS_LDS_POP) return st.finish_node_in_lds(tmax, key[0], ref[0], key[1], ref[1], key[2], ref[2], key[3], ref[3]);
        if (key[0] == 0xFFFFFFFFu) return pop_next<true>(st, tmax);
        if (key[3] != 0xFFFFFFFFu) st.template push<true>(ref[3], __uint_as_float(key[3]));
        if (key[2] != 0xFFFFFFFFu) st.template push<true>(ref[2], __uint_as_float(key[2]));
        if (key[1] != 0xFFFFFFFFu) st.template push<true>(ref[1], __uint_as_float(key[1]));
        return ref[0];
    }
    if (key[0] == 0xFFFFFFFFu) return pop_next(st, tmax);
    if (key[3] != 0xFFFFFFFFu) st.push(ref[3], __uint_as_float(key[3]));
    if (key[2] != 0xFFFFFFFFu) st.push(ref[2], __uint_as_float(key[2]));
    if (key[1] != 0xFFFFFFFFu) st.push(ref[1], __uint_as_float(key[1]));
    return ref[0];
}
#endif

// One leaf: primitive or medium test; a hit shrinks tmax.  Returns the next reference.
template <bool COUNT, bool GENERIC, class Stack, class Key>
RT_D uint32_t leaf_step(const DScene& sc, const TravRay& r, uint32_t cur, float& tmax, uint32_t& best, const Key& key, Stack& st,
                        Probe& pr) {
    float t;
    bool hit;
    if (ref_kind(cur) == KIND_MEDIUM) hit = medium_root<COUNT, GENERIC>(sc, ref_index(cur), r.o, r.d, r.time, r.a, r.tmin, tmax, key, t, pr);
    else hit = leaf_root<COUNT, true>(sc, cur, r.o, r.d, r.time, r.a, r.tmin, tmax, t, pr, r.inv);
    if (hit) {
        tmax = t;
        best = cur;
    }
    if (Stack::HAS_LDS_POP && st.wave_is_shallow(0)) return st.pop_next_in_lds(tmax, 0xFFFFFFFFu, 0u);
    return pop_next(st, tmax);
}

// The reference's box test of a sphere leaf in world space (AABB::from_points(c - r, c + r), src/hittable.rs:160-163,
// tested as src/math/aabb.rs:127-167 tests it).  When the origin lies inside the box, at least 2 tmin |d_k| from
// every face, the test passes whatever the rounding: each entry distance is negative, each exit distance is
// at least 2 tmin (1 - 2^-22) >= tmin (tmin >= 0), so the interval is [tmin, min(exits, tmax)] and not empty.  That case
// needs no division; every other ray takes the test itself.  try_inside: whether rays of this scene start inside the box
// as a rule (decided per scene: the attempt costs twenty instructions).
RT_D bool sphere_box_test(float4 s, const TravRay& r, float tmax, bool try_inside) {
    const float3 lo = f3(s.x - s.w, s.y - s.w, s.z - s.w), hi = f3(s.x + s.w, s.y + s.w, s.z + s.w);
    const float m = fminf(fminf(fminf(r.o.x - lo.x, hi.x - r.o.x), fminf(r.o.y - lo.y, hi.y - r.o.y)), fminf(r.o.z - lo.z, hi.z - r.o.z));
    const float two_tmin = 2.0f * r.tmin;
    if (try_inside && m > 0.0f && m >= two_tmin * fabsf(r.d.x) && m >= two_tmin * fabsf(r.d.y) && m >= two_tmin * fabsf(r.d.z) && 0.0f <= r.tmin && r.tmin <= tmax)
        return true;
    const float3 inv = f3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    float x0 = (lo.x - r.o.x) * inv.x, x1 = (hi.x - r.o.x) * inv.x;
    float y0 = (lo.y - r.o.y) * inv.y, y1 = (hi.y - r.o.y) * inv.y;
    float z0 = (lo.z - r.o.z) * inv.z, z1 = (hi.z - r.o.z) * inv.z;
    float t_in = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), r.tmin));
    float t_out = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
    return t_in <= t_out;
}

// Leaves whose boxes span (almost) the whole scene are kept out of the tree and tested first,
// by every lane at once.  A sphere still passes through the reference's box test of that leaf
// before Sphere::hit is evaluated.  A sphere followed by a medium bounded by that same sphere (a
// glass ball filled with fog, or the mist that fills a whole scene inside its boundary sphere) is
// one quadratic: both leaves take their roots from it, in the order and with the intervals they
// would have had one after the other.
// The pair "sphere leaf, then the medium that this same sphere bounds": one quadratic, whose two roots serve the sphere leaf
// and both boundary queries of the medium.
template <bool COUNT, class Key>
RT_D void sphere_and_its_medium(float4 s, const DMedium& m, uint32_t sphere_ref, uint32_t medium_ref, bool origin_inside, const TravRay& r,
                                float& tmax, uint32_t& best, const Key& key, Probe& pr) {
    if (COUNT) { pr.spheres += 3; pr.media++; }
    float near_root, far_root;
    if (sphere_roots(f3(s), s.w, r.o, r.d, r.a, near_root, far_root)) {
        float root = near_root;
        bool in = r.tmin < root && root < tmax;
        if (!in) {
            root = far_root;
            in = r.tmin < root && root < tmax;
        }
        if (in && sphere_box_test(s, r, tmax, origin_inside)) {
            tmax = root;
            best = sphere_ref;
        }
        float t1, t2, t;
        if (medium_span_of_roots(near_root, far_root, t1, t2) && medium_sample(m, ref_index(medium_ref), t1, t2, r.a, r.tmin, tmax, key, t)) {
            tmax = t;
            best = medium_ref;
        }
    }
}
RT_D bool medium_is_bounded_by(const DMedium& m, float4 s) {
    return m.single_sphere && m.sphere.x == s.x && m.sphere.y == s.y && m.sphere.z == s.z && m.sphere.w == s.w;
}

template <bool COUNT, bool GENERIC, class Key>
RT_D void global_leaves(const DScene& sc, const TravRay& r, float& tmax, uint32_t& best, const Key& key, Probe& pr) {
    for (uint32_t g = 0; g < sc.n_global; ++g) {
        uint32_t ref = sc.global_refs[g];
        float t;
        bool hit;
        if (ref_kind(ref) == KIND_MEDIUM) {
            hit = medium_root<COUNT, GENERIC>(sc, ref_index(ref), r.o, r.d, r.time, r.a, r.tmin, tmax, key, t, pr);
        } else {
            const float4 s = sc.spheres[ref_index(ref)];
            const uint32_t next = g + 1 < sc.n_global ? sc.global_refs[g + 1] : REF_NONE;
            if (next != REF_NONE && ref_kind(next) == KIND_MEDIUM) {
                const DMedium m = load_medium(sc.media + ref_index(next));
                if (medium_is_bounded_by(m, s)) {
                    sphere_and_its_medium<COUNT>(s, m, ref, next, (sc.global_inside >> g) & 1u, r, tmax, best, key, pr);
                    ++g;
                    continue;
                }
            }
            hit = sphere_box_test(s, r, tmax, (sc.global_inside >> g) & 1u) && leaf_root<COUNT>(sc, ref, r.o, r.d, r.time, r.a, r.tmin, tmax, t, pr);
        }
        if (hit) {
            tmax = t;
            best = ref;
        }
    }
}

// The one shape the reference's scenes have (a big sphere and the mist it bounds: the Book-2 scene) with its two records read
// once per thread, ahead of the rays: read inside a loop over rays or bounces they are scalar loads that the wave waits for, every
// time round.  Any other set of leaves goes through global_leaves.
struct FirstLeaves {
    bool pair = false;
    bool origin_inside = false;
    uint32_t ref0 = REF_NONE, ref1 = REF_NONE;
    float4 s;
    DMedium m;
};
RT_D FirstLeaves load_first_leaves(const DScene& sc) {
    FirstLeaves f;
    f.s = make_float4(0, 0, 0, 0);
    f.m = DMedium{};
    if (sc.n_global == 2) {
        f.ref0 = sc.global_refs[0];
        f.ref1 = sc.global_refs[1];
        if (ref_is_leaf(f.ref0) && ref_kind(f.ref0) == KIND_SPHERE && !ref_placed(f.ref0) && ref_kind(f.ref1) == KIND_MEDIUM) {
            f.s = sc.spheres[ref_index(f.ref0)];
            f.m = load_medium(sc.media + ref_index(f.ref1));
            f.pair = medium_is_bounded_by(f.m, f.s);
        }
    }
    f.origin_inside = (sc.global_inside & 1u) != 0;
    return f;
}
template <bool COUNT, bool GENERIC, class Key>
RT_D void first_leaves(const DScene& sc, const FirstLeaves& f, const TravRay& r, float& tmax, uint32_t& best, const Key& key, Probe& pr) {
    if (f.pair) sphere_and_its_medium<COUNT>(f.s, f.m, f.ref0, f.ref1, f.origin_inside, r, tmax, best, key, pr);
    else global_leaves<COUNT, GENERIC>(sc, r, tmax, best, key, pr);
}

// Starting point of a ray's traversal when the scene keeps leaves out of the tree: the closest
// hit among those leaves, as the hit record the extend kernel then improves on.  Ray generation
// computes it with the ray still in registers; for the rays shade creates, a streaming pass of
// its own (k_first_leaves) is faster than doing it inside the shade kernel's tile loop.
template <bool GENERIC>
RT_D uint2 first_leaves_hit(const DScene& sc, const BatchInfo& b, uint64_t seed, uint32_t bounce, float4 ro, float4 rd) {
    BatchMediumKey key{seed, __float_as_uint(rd.w), bounce, b};
    TravRay r = make_trav_ray(f3(ro), f3(rd), ro.w, RENDER_TMIN);
    float tmax = RT_INF;
    uint32_t best = REF_NONE;
    Probe pr;
    global_leaves<false, GENERIC>(sc, r, tmax, best, key, pr);
    return make_uint2(__float_as_uint(tmax), best);
}

// ---------------------------------------------------------------------------
// Closest hit over the flattened BVH (one ray per call): returns the leaf
// reference (REF_NONE on a miss) and its t.
// ---------------------------------------------------------------------------
// The tree walk of one ray from start to finish by one thread: improves (tmax, best).
template <bool COUNT, bool GENERIC, class Stack, class Key>
RT_D void walk_tree(const DScene& sc, const TravRay& r, float& tmax, uint32_t& best, const Key& key, Stack& st, Probe& pr) {
    const NodeSource src{sc.nodes, nullptr, 0};
    uint32_t cur = sc.root == REF_NONE ? REF_DONE : sc.root;
    // (the loop as one the wave leaves as a whole, its decisions made from counts: see the extend kernel's node loop)
    for (;;) {
        int n_inner = (int)__popcll(__ballot((int32_t)cur >= 0));  // inner nodes have no leaf bit; REF_DONE has
        int n_leaf = (int)__popcll(__ballot(cur < REF_DONE && ref_is_leaf(cur)));
        asm volatile("" : "+s"(n_inner), "+s"(n_leaf));
        if (n_inner + n_leaf == 0) break;
        if (n_inner != 0 && (int32_t)cur >= 0) cur = node_step<COUNT, false>(src, r, cur, tmax, st, pr);
        // the lanes at a leaf wait while at least half of the lanes that still walk are at an inner node: a leaf test runs to several
        // times the instructions of a node step (Book-1 configuration 2.945 -> 2.905 ms per call, 800x800 frame 1.392 -> 1.383 ms;
        // always waiting for the last inner lane: 2.95 and 1.43)
        if (n_inner != 0 && n_inner * 2 >= n_inner + n_leaf) continue;
        if (cur < REF_DONE && ref_is_leaf(cur)) cur = leaf_step<COUNT, GENERIC>(sc, r, cur, tmax, best, key, st, pr);
    }
}

template <bool COUNT>
RT_D uint32_t closest_hit(const DScene& sc, float3 o, float3 d, float time, float tmin, float tmax, const MediumKey& key, float& t_hit,
                          Probe& pr) {
    uint32_t best = REF_NONE;
    TravRay r = make_trav_ray(o, d, time, tmin);
    global_leaves<COUNT, true>(sc, r, tmax, best, key, pr);
    PrivateStack st;
    walk_tree<COUNT, true>(sc, r, tmax, best, key, st, pr);
    t_hit = tmax;
    return best;
}

// ---------------------------------------------------------------------------
// Hit geometry: HitRecord fields (src/hittable.rs:7-26) rebuilt from (ray, t, leaf)
// ---------------------------------------------------------------------------
struct HitGeom {
    float3 p, normal;
    float3 uv_dir;      // sphere: outward normal in object space (input of get_sphere_uv)
    float u, v;         // quad: alpha, beta; others filled on demand
    uint32_t material, prim_id;
    bool front_face;
    bool uv_ready;
};

// get_sphere_uv (src/hittable.rs:341-352)
RT_D void sphere_uv(float3 p, float& u, float& v) {
    const float PI = 3.14159265358979323846f;
    float theta = acosf(-p.y);
    float phi = atan2f(-p.z, p.x) + PI;
    u = phi / (2.0f * PI);
    v = theta / PI;
}

RT_D void set_face_normal(HitGeom& g, float3 d, float3 outward) {  // src/hittable.rs:22-25
    g.front_face = dot3(d, outward) < 0.0f;
    g.normal = g.front_face ? outward : -outward;
}

// m0: the first word group of the leaf's meta record (material, prim_id, transform, class).  Everything read here
// depends on the leaf reference only, so the loads go out together: the placement comes from the per-leaf copy.
RT_D HitGeom resolve_hit(const DScene& sc, uint32_t ref, const uint4& m0, float3 o, float3 d, float time, float t) {
    HitGeom g;
    g.u = 0.0f;
    g.v = 0.0f;
    g.uv_ready = true;
    g.uv_dir = f3(0, 0, 0);
    g.material = m0.x;
    g.prim_id = m0.y;
    uint32_t kind = ref_kind(ref), idx = ref_index(ref);
    if (kind == KIND_MEDIUM) {  // src/math/ray.rs:153-159
        g.p = o + t * d;
        g.normal = f3(1.0f, 0.0f, 0.0f);
        g.front_face = true;
        return g;
    }
    const bool placed = ref_placed(ref);
    float4 rot = make_float4(0, 0, 0, 0), off = rot;
    // (each kind works on its own copy of the ray, made after its loads have gone out: a copy made up here would make every
    // load below wait for the ray's direction to arrive)
    if (kind == KIND_QUAD) {
        const float4* q4 = sc.quads + 4 * (size_t)idx;
        float4 nd = get4(q4), qq = get4(q4 + 1), uu = get4(q4 + 2), vv = get4(q4 + 3);
        if (placed) { rot = get4(sc.quad_xf + 2 * (size_t)idx); off = get4(sc.quad_xf + 2 * (size_t)idx + 1); }
        float3 oq = o, dq = d;
        asm volatile("" : "+v"(oq.x), "+v"(oq.y), "+v"(oq.z), "+v"(dq.x), "+v"(dq.y), "+v"(dq.z));  // (the copy is made here, not earlier)
        if (placed) ray_to_object(rot, off, oq, dq);
        g.p = oq + t * dq;
        quad_alpha_beta(qq, uu, vv, g.p, g.u, g.v);
        set_face_normal(g, dq, f3(nd));
    } else {
        float3 c;
        float r;
        const float4* xf;
        if (kind == KIND_SPHERE) {
            float4 s = sc.spheres[idx];
            c = f3(s); r = s.w;
            xf = sc.sphere_xf + 2 * (size_t)idx;
        } else {
            float4 s0 = sc.moving[2 * (size_t)idx], s1 = sc.moving[2 * (size_t)idx + 1];
            c = f3(s0) + time * f3(s1); r = s0.w;
            xf = sc.moving_xf + 2 * (size_t)idx;
        }
        if (placed) { rot = get4(xf); off = get4(xf + 1); }
        float3 os = o, ds = d;
        asm volatile("" : "+v"(os.x), "+v"(os.y), "+v"(os.z), "+v"(ds.x), "+v"(ds.y), "+v"(ds.z));  // (the copy is made here, not earlier)
        if (placed) ray_to_object(rot, off, os, ds);
        g.p = os + t * ds;
        float3 outward = (1.0f / r) * (g.p - c);
        set_face_normal(g, ds, outward);
        g.uv_dir = outward;  // object space, as get_sphere_uv sees it (src/hittable.rs:284)
        g.uv_ready = false;
    }
    if (placed) hit_to_world(rot, off, g.p, g.normal);
    return g;
}

RT_D void ensure_uv(HitGeom& g) {
    if (!g.uv_ready) {
        sphere_uv(g.uv_dir, g.u, g.v);
        g.uv_ready = true;
    }
}

// ---------------------------------------------------------------------------
// PerlinNoise::noise / turbulance (src/perlin_noise.rs:39-102)
// ---------------------------------------------------------------------------
// Where the tables are read from: global memory, or the copy a block of the textured-shade kernel keeps in LDS
// (every lookup is a gather by 64 different indices; the LDS serves those several times faster than the L1 path).
typedef __attribute__((address_space(3))) const float4v lds_const_float4v;
typedef __attribute__((address_space(3))) const uint32_t lds_const_u32;
// (pointers that were themselves read from memory say "global" explicitly: the compiler cannot know, and would
// fall back to the slower flat loads)
typedef __attribute__((address_space(1))) const float4v global_const_float4v;
typedef __attribute__((address_space(1))) const uint32_t global_const_u32;
struct PerlinGlobal {
    const float4* vecs;
    const uint32_t* perm;
    RT_D float4 vec(uint32_t i) const { return f4(((global_const_float4v*)vecs)[i]); }
    RT_D uint32_t permuted(uint32_t i) const { return ((global_const_u32*)perm)[i]; }
};
struct PerlinLds {
    lds_const_float4v* vecs;
    lds_const_u32* perm;
    bool filled;  // false: no LDS copy, read the tables from global memory
    RT_D float4 vec(uint32_t i) const { return f4(vecs[i]); }
    RT_D uint32_t permuted(uint32_t i) const { return perm[i]; }
};
// (i & 255, (i + 1) & 255) for i = f as isize, f a whole number (src/perlin_noise.rs:47-49): the cast saturates at the ends of
// the 64-bit range and maps NaN to 0, the addition wraps.  From 2^31 on every f32 is a multiple of 256.
RT_D void lattice_bytes(float f, uint32_t (&b)[2]) {
    const bool small = fabsf(f) < 2147483648.0f;                 // false for NaN as well
    const bool top = f >= 9223372036854775808.0f;                // isize::MAX = ...ff, + 1 wraps to isize::MIN = ...00
    const int i = (int)(small ? f : 0.0f);
    b[0] = small ? (uint32_t)i & 255u : (top ? 255u : 0u);
    b[1] = small ? (uint32_t)(i + 1) & 255u : (top ? 0u : 1u);
}
template <class Tables>
RT_D float perlin_noise(const Tables& pn, float3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    uint32_t i[2], j[2], k[2];
    const bool far_out = !(fmaxf(fmaxf(fabsf(fx), fabsf(fy)), fabsf(fz)) < 2147483648.0f);   // or NaN
    if (__builtin_amdgcn_ballot_w64(far_out) == 0ull) {  // no lane of the wave is: the plain 32-bit conversion is in range
        const int ii = (int)fx, jj = (int)fy, kk = (int)fz;
        i[0] = (uint32_t)ii & 255u; i[1] = (uint32_t)(ii + 1) & 255u;
        j[0] = (uint32_t)jj & 255u; j[1] = (uint32_t)(jj + 1) & 255u;
        k[0] = (uint32_t)kk & 255u; k[1] = (uint32_t)(kk + 1) & 255u;
    } else {
        lattice_bytes(fx, i);
        lattice_bytes(fy, j);
        lattice_bytes(fz, k);
    }
    // the tables are expanded at scene creation (DPerlin), which takes the `% point_count` out of every lookup
    uint32_t ix[2] = {pn.permuted(i[0]), pn.permuted(i[1])};
    uint32_t iy[2] = {pn.permuted(256 + j[0]), pn.permuted(256 + j[1])};
    uint32_t iz[2] = {pn.permuted(512 + k[0]), pn.permuted(512 + k[1])};
    float uu = u * u * (3.0f - 2.0f * u);
    float vv = v * v * (3.0f - 2.0f * v);
    float ww = w * w * (3.0f - 2.0f * w);
    float accum = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float4 g = pn.vec(ix[a] ^ iy[b] ^ iz[c]);
                float3 wv = f3(u - (float)a, v - (float)b, w - (float)c);
                accum += (a ? uu : 1.0f - uu) * (b ? vv : 1.0f - vv) * (c ? ww : 1.0f - ww) * dot3(f3(g), wv);
            }
    return accum;
}

template <class Tables>
RT_D float perlin_turbulance(const Tables& pn, float3 p, int depth) {
    float accum = 0.0f, weight = 1.0f;
#pragma unroll 1
    for (int i = 0; i < depth; ++i) {
        accum += weight * perlin_noise(pn, p);
        weight *= 0.5f;
        p = 2.0f * p;
    }
    return fabsf(accum);
}

// What Material::{emitted, scatter} need to know about a hit's material: kind, parameter and
// its texture (the colour itself when the texture is a solid colour).
struct ShadeRec {
    uint32_t kind, tex_kind, tex_id;
    float param;
    float3 colour;
};

// From the material table (two dependent reads): entry points that are handed a MaterialId.
RT_D ShadeRec shade_rec_of_material(const DScene& sc, uint32_t material) {
    DMaterial m = sc.materials[material];
    DTexture t = sc.textures[m.texture];
    ShadeRec r;
    r.kind = m.kind; r.param = m.param; r.tex_kind = t.kind; r.tex_id = m.texture; r.colour = f3(t.r, t.g, t.b);
    return r;
}

// From the leaf's own 32-byte record (one read): the render path.
RT_D const uint4* leaf_meta(const DScene& sc, uint32_t ref) {
    size_t i = 2 * (size_t)ref_index(ref);
    switch (ref_kind(ref)) {
        case KIND_SPHERE: return sc.sphere_meta + i;
        case KIND_MOVING: return sc.moving_meta + i;
        case KIND_QUAD: return sc.quad_meta + i;
        default: return sc.medium_meta + i;
    }
}
RT_D ShadeRec shade_rec_of_leaf(const DScene& sc, uint4 m0, uint4 m1) {
    ShadeRec r;
    r.kind = (m0.w >> 8) & 0xFFu;
    r.tex_kind = (m0.w >> 16) & 0xFFu;
    r.param = __uint_as_float(m1.x);
    r.colour = f3(__uint_as_float(m1.y), __uint_as_float(m1.z), __uint_as_float(m1.w));
    r.tex_id = 0;
    if (r.tex_kind != RT_TEX_SOLID) r.tex_id = sc.materials[m0.x].texture;
    return r;
}

// Rust's `as i32` / `as u32`: saturating, NaN becomes 0 (a C cast of such values is undefined)
RT_D int rust_as_i32(float f) { return f != f ? 0 : (f >= 2147483648.0f ? 2147483647 : (f <= -2147483648.0f ? (-2147483647 - 1) : (int)f)); }
RT_D uint32_t rust_as_u32(float f) { return !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f); }
// is_even of Texture::Checker (src/texture.rs:46-48): the floor of all FOUR lanes of inv_scale * p is summed, and the fourth
// lane is inv_scale * 0 - zero, except for a checker of scale 0 (inv_scale = inf), where it is NaN and the sum with it.
RT_D bool checker_is_even(float inv_scale, float3 p) {
    const float sum = ((floorf(inv_scale * p.x) + floorf(inv_scale * p.y)) + floorf(inv_scale * p.z)) + floorf(inv_scale * 0.0f);
    return (rust_as_i32(sum) % 2) == 0;
}

// Texture::value (src/texture.rs:40-73) for textures that are not a plain colour
RT_D float3 texture_value_slow(const DScene& sc, uint32_t tex_id, HitGeom& g, PerlinLds lds_perlin) {
    DTexture t = sc.textures[tex_id];
    for (int guard = 0; guard < 16 && t.kind == RT_TEX_CHECKER; ++guard) {
        t = sc.textures[checker_is_even(t.scale, g.p) ? t.even : t.odd];
    }
    if (t.kind == RT_TEX_SOLID) return f3(t.r, t.g, t.b);
    if (t.kind == RT_TEX_IMAGE) {
        ensure_uv(g);
        DImage im = sc.images[t.image];
        // f32::clamp hands a NaN through (v = acos(-n.y) / pi is NaN where rounding leaves |n.y| a hair above 1), `as u32` turns it into 0
        float u = g.u < 0.0f ? 0.0f : (g.u > 1.0f ? 1.0f : g.u);
        float v = 1.0f - (g.v < 0.0f ? 0.0f : (g.v > 1.0f ? 1.0f : g.v));
        uint32_t i = rust_as_u32(u * (float)(im.width - 1));
        uint32_t j = rust_as_u32(v * (float)(im.height - 1));
        size_t at = (size_t)j * im.width + i;
        float3 px;
        if (im.format == RT_IMAGE_RGB8) {
            const uint32_t c = ((global_const_u32*)im.texels)[at];  // uchar4: r, g, b, unused
            px = f3((float)(c & 0xFFu) / 255.0f, (float)((c >> 8) & 0xFFu) / 255.0f, (float)((c >> 16) & 0xFFu) / 255.0f);
        } else {
            px = f3(f4(((global_const_float4v*)im.texels)[at]));
        }
        return px * px;
    }
    if (t.kind == RT_TEX_NOISE) {
        float turb;
        if (lds_perlin.filled) {  // the scene's one table, copied to LDS by the caller
            turb = perlin_turbulance(lds_perlin, g.p, 7);
        } else {
            const DPerlin pn = sc.perlins[t.perlin];
            turb = perlin_turbulance(PerlinGlobal{pn.vecs, pn.perm}, g.p, 7);
        }
        float s = 1.0f + sinf(t.scale * g.p.z + 10.0f * turb);
        return f3(s * 0.5f, s * 0.5f, s * 0.5f);
    }
    return f3(0, 0, 0);
}

template <bool TEXTURES>
RT_D float3 texture_value(const DScene& sc, const ShadeRec& rec, HitGeom& g, PerlinLds lds_perlin) {
    if (!TEXTURES || rec.tex_kind == RT_TEX_SOLID) return rec.colour;
    return texture_value_slow(sc, rec.tex_id, g, lds_perlin);
}

// Vec3::random_unit (src/math/vec3.rs:57-67)
RT_D float3 random_unit(Rng& rng) {
    for (;;) {
        float a = -1.0f + 2.0f * rng.next();
        float b = -1.0f + 2.0f * rng.next();
        float c = -1.0f + 2.0f * rng.next();
        float3 p = f3(a, b, c);
        if (dot3(p, p) < 1.0f) return unit3(p);
    }
}

// reflectance (src/material.rs:155-160)
RT_D float schlick(float cosine, float rr) {
    float r0 = (1.0f - rr) / (1.0f + rr);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    return r0 + (1.0f - r0) * (x * (x2 * x2));
}

struct ScatterResult {
    float3 dir, attenuation, emitted;
    bool scattered;
};

// Material::emitted + Material::scatter (src/material.rs:65-122).  Every kind evaluates its
// texture exactly once (attenuation, or the emitted colour of a light), so that is done first.
template <bool TEXTURES>
RT_D ScatterResult shade_hit(const DScene& sc, const ShadeRec& m, HitGeom& g, float3 d_in, Rng& rng, PerlinLds lds_perlin = PerlinLds{nullptr, nullptr, false}) {
    ScatterResult r;
    r.emitted = f3(0, 0, 0);
    r.attenuation = f3(0, 0, 0);
    r.dir = f3(0, 0, 0);
    r.scattered = false;
    float3 tex = texture_value<TEXTURES>(sc, m, g, lds_perlin);
    switch (m.kind) {
        case RT_MAT_LAMBERTIAN: {
            float3 dir = g.normal + random_unit(rng);
            if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = g.normal;
            r.dir = dir;
            r.attenuation = tex;
            r.scattered = true;
            break;
        }
        case RT_MAT_METAL: {
            float fuzz = fminf(m.param, 1.0f);
            float3 ud = unit3(d_in);
            float3 reflected = ud - (2.0f * dot3(ud, g.normal)) * g.normal;
            float3 dir = reflected + fuzz * random_unit(rng);
            if (dot3(dir, g.normal) > 0.0f) {
                r.dir = dir;
                r.attenuation = tex;
                r.scattered = true;
            }
            break;
        }
        case RT_MAT_DIELECTRIC: {
            r.attenuation = tex;
            float ratio = g.front_face ? 1.0f / m.param : m.param;
            float3 ud = unit3(d_in);
            float cos_theta = fminf(dot3(-ud, g.normal), 1.0f);
            float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            bool cannot_refract = ratio * sin_theta > 1.0f;
            // the uniform draw happens only when refraction is possible (short-circuit at :97)
            if (cannot_refract || schlick(cos_theta, ratio) > rng.next()) {
                r.dir = ud - (2.0f * dot3(ud, g.normal)) * g.normal;
            } else {  // Vec3::refract, src/math/vec3.rs:89-94
                float3 perp = ratio * (ud + cos_theta * g.normal);
                float3 par = (-sqrtf(fabsf(1.0f - dot3(perp, perp)))) * g.normal;
                r.dir = perp + par;
            }
            r.scattered = true;
            break;
        }
        case RT_MAT_ISOTROPIC: {
            r.dir = random_unit(rng);
            r.attenuation = tex;
            r.scattered = true;
            break;
        }
        case RT_MAT_DIFFUSE_LIGHT: {
            r.emitted = tex;
            break;
        }
        default: break;
    }
    return r;
}

// ---------------------------------------------------------------------------
// counters
// ---------------------------------------------------------------------------
#ifndef RT_LANE_STATS
#define RT_LANE_STATS 0  // development builds: 1..8 count the lane slots of one loop part of k_extend in DCounters::max_stack
#endif
RT_D unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

RT_D void flush_probe(DCounters* c, const Probe& pr, uint32_t segments, uint32_t hits, uint32_t slots = 0) {
#if RT_LANE_STATS != 0
    if (slots) {  // development builds reuse the field for the lane-slot count of one loop part
        unsigned long long sl = wave_sum(slots);
        if ((threadIdx.x & 63) == 0) atomicAdd(&c->max_stack, sl);
    }
#else
    {
        uint32_t deepest = pr.max_sp;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) deepest = max(deepest, (uint32_t)__shfl_down((int)deepest, off, 64));
        if ((threadIdx.x & 63) == 0 && deepest != 0) atomicMax(&c->max_stack, (unsigned long long)deepest);
    }
#endif
    unsigned long long n = wave_sum(pr.nodes), s = wave_sum(pr.spheres), q = wave_sum(pr.quads), m = wave_sum(pr.media);
    unsigned long long sg = wave_sum(segments), h = wave_sum(hits);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&c->node_visits, n);
        atomicAdd(&c->sphere_tests, s);
        atomicAdd(&c->quad_tests, q);
        atomicAdd(&c->medium_tests, m);
        atomicAdd(&c->segments, sg);
        atomicAdd(&c->hits, h);
    }
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

// The throughput queue holds three floats per slot (12-byte stride).
RT_D float4 load_thr(const float* base, uint32_t i) {
    const float* p = base + 3 * (size_t)i;
    return make_float4(p[0], p[1], p[2], 0.0f);
}
RT_D void store_thr(float* base, uint32_t i, float4 v) {
    float* p = base + 3 * (size_t)i;
    p[0] = v.x; p[1] = v.y; p[2] = v.z;
}
// dst = v as one 16-byte store of the four components.  (Assigned as structs, a path's radiance and its next ray are written
// component by component; the compiler then merges the component stores of the branches that end a path with those of the
// branch that continues it into one block with a pointer that is either - and the next ray lives in scratch memory, written
// through flat stores.)
RT_D void put4(float4* dst, const float4& v) { *(float4n*)dst = float4n{v.x, v.y, v.z, v.w}; }

// get_ray + pixel_sample_square + defocus_disk_sample (src/camera.rs:311-320,368-380).
// Draw order on the camera stream: jitter x, jitter y, [disk trials], time.
RT_D void primary_ray(const DCamera& cam, uint64_t seed, uint32_t pixel, uint32_t sample, float3& origin, float3& dir, float& time) {
    uint32_t x = pixel % cam.width, y = pixel / cam.width;
    Rng rng;
    rng.init(path_key(seed, pixel, sample), 0, STREAM_CAMERA);
    float3 centre = cam.pixel00 + ((float)x * cam.du) + ((float)y * cam.dv);
    float px = -0.5f + rng.next();
    float py = -0.5f + rng.next();
    float3 sample_pt = centre + (px * cam.du + py * cam.dv);
    origin = cam.centre;
    if (cam.defocus_angle > 0.0f) {
        float dx, dy;
        do {
            dx = -1.0f + 2.0f * rng.next();
            dy = -1.0f + 2.0f * rng.next();
        } while (!(dx * dx + dy * dy < 1.0f));
        origin = cam.centre + (dx * cam.ddu) + (dy * cam.ddv);
    }
    dir = sample_pt - origin;
    time = rng.next();
}

// One thread per path of the batch.  s_local * n_pixels + pixel_local is the path's initial queue slot and its
// slot in the radiance array.
template <bool GENERIC>
__global__ __launch_bounds__(256) void k_raygen(DScene sc, DCamera cam, BatchInfo b, uint64_t seed, float4* __restrict__ ray_o,
                                                float4* __restrict__ ray_d, float* __restrict__ thr, uint2* __restrict__ first_hits,
                                                uint32_t* __restrict__ counts, uint32_t n_counts) {
    const uint32_t total = b.n_pixels * b.n_samples;
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < n_counts; i += blockDim.x) counts[i] = (i == 0) ? total : 0u;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint32_t s_local = i / b.n_pixels;
        uint32_t pixel = b.first_pixel + (i - s_local * b.n_pixels);
        float3 origin, dir;
        float time;
        primary_ray(cam, seed, pixel, b.first_sample + s_local, origin, dir, time);
        const uint32_t id = (s_local << b.pixel_bits) | (i - s_local * b.n_pixels);
        const float4 ro = make_float4(origin.x, origin.y, origin.z, time), rd = make_float4(dir.x, dir.y, dir.z, __uint_as_float(id));
        ray_o[i] = ro;
        ray_d[i] = rd;
        store_thr(thr, i, make_float4(1.0f, 1.0f, 1.0f, 0.0f));
        if (first_hits) first_hits[i] = first_leaves_hit<GENERIC>(sc, b, seed, 0, ro, rd);
    }
}

__global__ __launch_bounds__(256) void k_primary_rays(DCamera cam, uint64_t seed, uint32_t sample, const uint32_t* __restrict__ pixels,
                                                      unsigned long long n, rt_ray* __restrict__ out) {
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        float3 origin, dir;
        float time;
        primary_ray(cam, seed, pixels[i], sample, origin, dir, time);
        rt_ray r = rt_ray{};
        r.origin[0] = origin.x; r.origin[1] = origin.y; r.origin[2] = origin.z; r.time = time;
        r.direction[0] = dir.x; r.direction[1] = dir.y; r.direction[2] = dir.z;
        r.tmin = 0.001f; r.tmax = RT_INF; r.rng_pixel = pixels[i]; r.rng_sample = sample;
        out[i] = r;
    }
}

// Extend: closest hit for every ray of the current queue.
//
// Persistent waves: wave w owns every n_waves-th 64-ray block of the queue and its 64 lanes
// pull rays from them one by one, so a lane that finishes a short ray starts the next
// instead of idling until the longest ray of a 64-ray batch is done.
//
// Each turn of the loop has four parts, each run by the lanes it concerns: fetching new rays
// (once REFILL_MIN lanes are free), stepping through inner nodes (while INNER_MIN lanes are at
// one), testing the sphere / quad leaves lanes have arrived at, and testing media.  The medium
// test is by far the longest, so lanes that arrive at a medium wait there until MEDIUM_MIN of
// them have gathered (or nothing else is left to do).
#ifndef RT_EXTEND_BLOCK
#define RT_EXTEND_BLOCK 256
#endif
static constexpr int EXTEND_BLOCK = RT_EXTEND_BLOCK;

// LDS of one block: 22 KiB (seven blocks to a CU), spent either on 11 stack levels per lane, or - for a tree small
// enough to be copied into LDS whole - on 8 stack levels and the tree (a copy of only the top
// levels of a larger tree does not pay: those nodes are served by the L1 cache anyway, and
// more LDS per block means fewer resident waves).
static constexpr int SMALL_TREE_NODES = 96;  // 6 KiB of nodes
#ifndef RT_EXTEND_STACK
#define RT_EXTEND_STACK 11  // stack levels per lane kept in LDS (large trees): 22 KiB per block, seven blocks per CU
#endif
#ifndef RT_EXTEND_TOP_NODES
#define RT_EXTEND_TOP_NODES 0
#endif
template <bool SMALL_TREE> struct ExtendLds {
    static constexpr int STACK_LEVELS = SMALL_TREE ? 8 : RT_EXTEND_STACK;
    static constexpr int TOP_NODES = SMALL_TREE ? SMALL_TREE_NODES : RT_EXTEND_TOP_NODES;
};
static constexpr int MIN_LDS_STACK = 8;
static constexpr int OVERFLOW_STACK = STACK_SIZE - MIN_LDS_STACK;
#ifndef RT_EXTEND_WAVES
#define RT_EXTEND_WAVES 7  // resident waves per SIMD the register allocation aims for (72 registers)
#endif
#ifndef RT_EXTEND_WAVES_GENERIC
#define RT_EXTEND_WAVES_GENERIC 7  // the same for scenes whose media have boxes or hulls as boundaries: 71 registers, a few
                                   // scalar registers kept in spare vector lanes (Cornell smoke 1024x1024x16: 12.4 ms at five waves, 11.5 at seven)
#endif
template <bool COUNT, bool SMALL_TREE, bool GENERIC, bool OVERFLOW = true>
__global__ __launch_bounds__(EXTEND_BLOCK) __attribute__((amdgpu_waves_per_eu(GENERIC ? RT_EXTEND_WAVES_GENERIC : RT_EXTEND_WAVES, GENERIC ? RT_EXTEND_WAVES_GENERIC : RT_EXTEND_WAVES))) void k_extend(DScene sc, BatchInfo b, uint64_t seed, uint32_t bounce,
                                                         const uint32_t* __restrict__ counts, const float4* __restrict__ ray_o,
                                                         const float4* __restrict__ ray_d, uint2* __restrict__ hits,
                                                         unsigned long long* __restrict__ stack_overflow, ExtendTuning tune,
                                                         DCounters* counters) {
    constexpr int LDS_STACK = ExtendLds<SMALL_TREE>::STACK_LEVELS, LDS_TOP_NODES = ExtendLds<SMALL_TREE>::TOP_NODES;
    __shared__ unsigned long long s_stack[EXTEND_BLOCK / 64][LDS_STACK][64];
    __shared__ float4 s_top[LDS_TOP_NODES > 0 ? NODE_FLOAT4S * LDS_TOP_NODES : 1];
    NodeSource src{sc.nodes, s_top, 0};
    if (LDS_TOP_NODES > 0) {  // the block's copy of the tree
        src.n_top = min(sc.n_nodes, (uint32_t)LDS_TOP_NODES);
        for (uint32_t i = threadIdx.x; i < NODE_FLOAT4S * src.n_top; i += EXTEND_BLOCK) s_top[i] = sc.nodes[i];
        __syncthreads();
    }
    __shared__ uint32_t s_cursor;  // 64-ray blocks the waves of this thread block have taken so far
    if (threadIdx.x == 0) s_cursor = 0;
    __syncthreads();
    // (values that are the same in all lanes of a wave are said to be so explicitly: that keeps the queue bookkeeping and the
    // loop's decisions below in scalar registers and scalar branches)
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)counts[bounce]);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wave = blockIdx.x * (EXTEND_BLOCK / 64) + wave_in_block, n_waves = gridDim.x * (EXTEND_BLOCK / 64);
    // A thread block owns EXTEND_BLOCK / 64 consecutive 64-ray blocks out of every n_waves of the queue; its waves take them
    // one at a time from a counter in LDS, so the waves of a thread block finish together (the sum of several waves' work
    // varies less from thread block to thread block than one wave's does from wave to wave).
    auto take_block = [&]() -> uint32_t {
        uint32_t i = 0;
        if (lane == 0) i = __hip_atomic_fetch_add(&s_cursor, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        constexpr uint32_t W = EXTEND_BLOCK / 64;
        return (i / W) * n_waves + blockIdx.x * W + (i % W);
    };
    // Work of this wave: the 64-ray blocks w, w + n_waves, w + 2 n_waves, ... of the queue.  Interleaving
    // spreads expensive image regions over all waves (a contiguous slice per wave does not: the rows of
    // an image differ a lot in cost), with no atomics.
    uint32_t blk = take_block();
    if ((unsigned long long)blk * 64ull >= n) return;
    uint32_t next = blk * 64u;
    uint32_t blk_end = min(n, next + 64u);

    WaveStack<LDS_STACK, OVERFLOW> st;
    st.lds = (lds_u64*)&s_stack[wave_in_block][0][lane];
    st.ovf = (global_u64*)(stack_overflow + ((size_t)wave * OVERFLOW_STACK) * 64);
    st.lane = lane;
    Probe pr;
    uint32_t segs = 0, nhit = 0, slots = 0;

    // State of a lane: cur is REF_NONE while the lane has no ray, REF_DONE when its ray is finished and waits to be written
    // back, else the reference the ray is at.  Every question the loop asks is then one comparison on cur: inner nodes are
    // below 2^31, sphere / moving-sphere / quad leaves 0x80000000 .. 0xDFFFFFFF, media 0xE0000000 .. 0xEFFFFFFF (a medium's
    // reference never has the placed bit), REF_DONE and REF_NONE at the very top.
    static_assert((REF_LEAF | (KIND_MEDIUM << 29)) == 0xE0000000u && REF_PLACED == 0x10000000u && KIND_MEDIUM == 3u && REF_DONE >= 0xF0000000u,
                  "the comparisons below spell out the layout of a reference");
    auto is_inner = [](uint32_t c) { return (int32_t)c >= 0; };
    auto is_sphere = [](uint32_t c) { return (int32_t)c < (int32_t)0xC0000000u; };  // still or moving
    auto is_quad = [](uint32_t c) { return c - 0xC0000000u < 0x20000000u; };
    auto is_medium = [](uint32_t c) { return c - 0xE0000000u < 0x10000000u; };
    uint32_t mine = 0;  // queue slot this lane is working on (meaningful while cur != REF_NONE)
    uint32_t cur = REF_NONE, best = REF_NONE;
    float tmax = RT_INF;
    TravRay r;
    BatchMediumKey key{seed, 0, bounce, b};
    const uint32_t root = sc.root == REF_NONE ? REF_DONE : sc.root;
    const bool first_leaves_done = sc.n_global != 0;  // k_first_leaves has run: hits[] holds each ray's starting (tmax, leaf)

    // A turn in which no part ran (the thresholds below held every lane back) is followed by one
    // that runs every part for whatever lanes it has, so the loop always moves on.
    bool eager = false;
#ifdef RT_WAVE_TIMES  // development aid: prints the life span and the node-loop turns of every 16th wave
    const unsigned long long t_start = wall_clock64();
    uint32_t node_turns = 0, overflow_turns = 0, my_visits = 0;
#endif
#ifdef RT_SECTION_TIMES  // development aid: shader-clock cycles every 64th wave spends in each part of the loop
    unsigned long long t_sect[5] = {0, 0, 0, 0, 0}, t_mark = __builtin_readcyclecounter();
    uint32_t n_sect[5] = {0, 0, 0, 0, 0};
    const unsigned long long t_begin = t_mark;
#define RT_SECTION_END(i, ran) { const unsigned long long t_now = __builtin_readcyclecounter(); t_sect[i] += t_now - t_mark; n_sect[i] += (ran) ? 1u : 0u; t_mark = t_now; }
#else
#define RT_SECTION_END(i, ran)
#endif
    for (;;) {
        bool moved = false;
        const int refill_min = eager ? 1 : tune.refill_min, inner_min = eager ? 1 : tune.inner_min, medium_min = eager ? 1 : tune.medium_min,
                  quad_min = eager ? 1 : tune.quad_min;
        if (__ballot(cur == REF_DONE)) moved = true;
        if (cur == REF_DONE) {  // retire
#ifdef RT_WAVE_TIMES
            if (my_visits > 3000) printf("ray %u visits %u o %.9g %.9g %.9g d %.9g %.9g %.9g t %g best %x\n", mine, my_visits, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, tmax, best);
            my_visits = 0;
#endif
            hits[mine] = make_uint2(__float_as_uint(tmax), best);
            if (COUNT) { segs++; nhit += (best != REF_NONE); }
            cur = REF_NONE;
        }
        if (next == blk_end && next < n) {  // block used up: on to this wave's next one
            blk = take_block();
            unsigned long long nb = (unsigned long long)blk * 64ull;
            next = (uint32_t)min(nb, (unsigned long long)n);
            blk_end = (uint32_t)min(nb + 64ull, (unsigned long long)n);
        }
        const uint32_t avail = blk_end - next;
        const unsigned long long m_idle = __ballot(cur == REF_NONE);
        if (m_idle == ~0ull && avail == 0) break;
        const uint32_t n_refill = min((uint32_t)__popcll(m_idle), avail);
#if RT_LANE_STATS == 4
        if (COUNT) slots++;
#endif
        unsigned long long m_busy = __ballot(cur < 0xC0000000u);  // at an inner node or a sphere: lanes whose work does not wait for others
        if (n_refill != 0 && ((int)__popcll(m_idle) >= refill_min || m_busy == 0)) {
            moved = true;
#if RT_LANE_STATS == 3
            if (COUNT) slots++;
#endif
            uint32_t rank = (uint32_t)__popcll(m_idle & ((1ull << lane) - 1ull));
            const bool fetch = cur == REF_NONE && rank < avail;
            if (fetch) {
                mine = next + rank;
                // the ray and its starting hit record in one round trip: all three loads go out before anything waits
                const float4 ro = get4(ray_o + mine), rd = get4(ray_d + mine);
                uint2 h0 = make_uint2(__float_as_uint(RT_INF), REF_NONE);
                if (first_leaves_done) h0 = hits[mine];
                asm volatile("" : "+v"(h0.x), "+v"(h0.y));  // (keeps the compiler from moving that load below the divisions)
                key.path = __float_as_uint(rd.w);
                r = make_trav_ray(f3(ro), f3(rd), ro.w, RENDER_TMIN);
                tmax = __uint_as_float(h0.x);
                best = h0.y;
                st.sp = 0;
                cur = root;
            }
            next += n_refill;
        }
        RT_SECTION_END(0, true)
        // inner nodes: as long as enough lanes are at one, or the lanes that are have nothing else to wait for.  (The decision
        // is the same for the whole wave and is computed as one scalar value, so that the loop is a scalar loop: left as a
        // loop with several exits, each costs every turn a handful of exec-mask instructions.)
        auto node_turn_due = [&]() -> bool {
            // (counts, not masks compared with zero: the compiler keeps a test of a ballot against zero as a per-lane
            // condition, and the loop would be one that each lane leaves for itself)
            const int n_inner = (int)__popcll(__ballot(is_inner(cur)));
            int due = n_inner;
            if (n_inner != 0 && n_inner < inner_min) {
                const int n_sphere = (int)__popcll(__ballot(is_sphere(cur))), n_quad = (int)__popcll(__ballot(is_quad(cur))),
                          n_med = (int)__popcll(__ballot(is_medium(cur)));
                if (n_sphere != 0 || n_quad >= quad_min || n_med >= medium_min) due = 0;
                else if (next < n && (int)__popcll(__ballot(cur >= REF_DONE)) >= refill_min) due = 0;  // lanes free or about to be
            }
            asm volatile("" : "+s"(due));  // (a value in a scalar register is the same in every lane by construction)
            return due != 0;
        };
        while (node_turn_due()) {
            const bool at_inner = is_inner(cur);
            moved = true;
#if RT_LANE_STATS == 1
            if (COUNT) slots++;
#endif
            if (at_inner) {
                cur = node_step<COUNT, (LDS_TOP_NODES > 0)>(src, r, cur, tmax, st, pr);
                if (COUNT) pr.max_sp = max(pr.max_sp, (uint32_t)st.sp);
#if RT_LANE_STATS == 7
                if (COUNT && st.sp > LDS_STACK) slots++;  // node steps that leave the lane's stack in the global part
#endif
            }
#if RT_LANE_STATS == 8
            if (COUNT && __ballot(at_inner && st.sp > LDS_STACK)) slots++;  // turns of the node loop with some lane's stack in the global part
#endif
#ifdef RT_WAVE_TIMES
            if (at_inner) my_visits++;
            node_turns++;
            if (__ballot(at_inner && st.sp > LDS_STACK)) overflow_turns++;
#endif
            RT_SECTION_END(1, true)
        }
        RT_SECTION_END(0, false)
        // sphere leaves
        {
            const bool at_sphere = is_sphere(cur);
            const bool shallow = st.wave_is_shallow(0);  // (no lane's stack reaches into the global part: pops read LDS)
#if RT_LANE_STATS == 2
            if (COUNT && __ballot(at_sphere)) slots++;
#endif
            if (__ballot(at_sphere)) moved = true;
            if (at_sphere) {
                float t;
                if (sphere_leaf_root<COUNT, true>(sc, cur, r.o, r.d, r.time, r.a, r.tmin, tmax, t, pr, r.inv)) { tmax = t; best = cur; }
                cur = shallow ? st.pop_next_in_lds(tmax, 0xFFFFFFFFu, 0u) : pop_next(st, tmax);
            }
            RT_SECTION_END(2, __ballot(at_sphere) != 0)
        }
        // quad leaves (and media, below): once enough lanes wait at one, or when the wave has nothing else it could do.  A quad's
        // test is two memory round trips and as many instructions as a sphere's; run for every turn's two or three quads it
        // would double the cost of the leaf part.
        const bool can_refill = avail > n_refill || (__ballot(cur == REF_NONE) != 0 && next < n);
        {
            const bool at_quad = is_quad(cur);
            const unsigned long long m_quad = __ballot(at_quad);
            bool ran = false;
            if (m_quad != 0) {
                bool run = (int)__popcll(m_quad) >= quad_min;
                if (!run) run = __ballot(cur < 0xC0000000u) == 0 && !can_refill;  // no lane at an inner node or a sphere, none to fill
                if (run) {
                    moved = ran = true;
                    const bool shallow = st.wave_is_shallow(0);
                    if (at_quad) {
                        float t;
                        if (quad_leaf_root<COUNT, true>(sc, cur, r.o, r.d, r.tmin, tmax, t, pr, r.inv)) { tmax = t; best = cur; }
                        cur = shallow ? st.pop_next_in_lds(tmax, 0xFFFFFFFFu, 0u) : pop_next(st, tmax);
                    }
                }
            }
            RT_SECTION_END(3, ran)
        }
        // media: once enough lanes wait at one, or when the wave has nothing else it could do
        {
            const bool at_medium = is_medium(cur);
            const unsigned long long m_med = __ballot(at_medium);
            if (m_med != 0) {
                bool run = (int)__popcll(m_med) >= medium_min;
                if (!run) {
                    const unsigned long long m_other = __ballot(cur != REF_NONE) & ~m_med;  // includes finished rays: they free lanes
                    run = m_other == 0 && !can_refill;
                }
                if (run) {
                    moved = true;
#if RT_LANE_STATS == 6
                    if (COUNT) slots++;
#endif
                    if (at_medium) {
                        float t;
                        if (medium_root<COUNT, GENERIC>(sc, ref_index(cur), r.o, r.d, r.time, r.a, r.tmin, tmax, key, t, pr)) { tmax = t; best = cur; }
                        cur = pop_next(st, tmax);
                    }
                }
            }
        }
        RT_SECTION_END(4, __ballot(is_medium(cur)) != 0)
        eager = __ballot(!moved) != 0ull;
    }
#ifdef RT_SECTION_TIMES
    if (lane == 0 && (wave & 63u) == 0)
        printf("sections %u %u total %llu fetch %llu %u node %llu %u sphere %llu %u quad %llu %u medium %llu %u\n", bounce, wave, __builtin_readcyclecounter() - t_begin,
               t_sect[0], n_sect[0], t_sect[1], n_sect[1], t_sect[2], n_sect[2], t_sect[3], n_sect[3], t_sect[4], n_sect[4]);
#endif
#ifdef RT_WAVE_TIMES
    if (lane == 0 && ((wave & 255u) == 0 || node_turns > RT_WAVE_TIMES)) printf("wave %u %u start %llu end %llu turns %u overflow %u\n", bounce, wave, t_start, wall_clock64(), node_turns, overflow_turns);
#endif
    if (COUNT) flush_probe(counters, pr, segs, nhit, slots);
}

// Runs ahead of k_extend when the scene has leaves that are kept out of the tree: tests them
// for every ray of the queue with all lanes busy, and leaves the closest hit so far in hits[].
template <bool COUNT, bool GENERIC>
__global__ __launch_bounds__(256) void k_first_leaves(DScene sc, BatchInfo b, uint64_t seed, uint32_t bounce,
                                                      const uint32_t* __restrict__ counts, const float4* __restrict__ ray_o,
                                                      const float4* __restrict__ ray_d, uint2* __restrict__ hits, DCounters* counters) {
    const uint32_t n = counts[bounce];
    Probe pr;
    const FirstLeaves fl = load_first_leaves(sc);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 ro = ray_o[i], rd = ray_d[i];
        BatchMediumKey key{seed, __float_as_uint(rd.w), bounce, b};
        TravRay r = make_trav_ray(f3(ro), f3(rd), ro.w, RENDER_TMIN);
        float tmax = RT_INF;
        uint32_t best = REF_NONE;
        first_leaves<COUNT, GENERIC>(sc, fl, r, tmax, best, key, pr);
        hits[i] = make_uint2(__float_as_uint(tmax), best);
    }
    if (COUNT) flush_probe(counters, pr, 0, 0);
}

// One (ray, hit) pair through ray_colour's loop body (src/camera.rs:344-361): writes the path's
// radiance when the path ends here, otherwise returns the next ray.
// TEXTURES: image / noise / nested checker textures are evaluated here (k_shade leaves those hits to k_shade_textured);
// PLAIN_CHECKERS: a checker of two plain colours is evaluated from its per-material record.
template <bool TEXTURES, bool PLAIN_CHECKERS = !TEXTURES>
RT_D bool shade_one(const DScene& sc, const DCamera& cam, const BatchInfo& b, uint64_t seed, uint32_t bounce, float4 ro, float4 rd, float4 th,
                    uint2 h, float4* __restrict__ radiance, float4& no, float4& nd, float4& nt, PerlinLds lds_perlin = PerlinLds{nullptr, nullptr, false}) {
    // The path's id, (sample of the batch << pixel_bits) | pixel of the batch, travels in the direction's fourth lane.  What
    // is derived from it is worked out where it is used and not up here: this way nothing waits for the direction before the
    // loads of the leaf's records (which only need the reference) have gone out - one memory round trip instead of two.
    auto s_local_of = [&]() { return __float_as_uint(rd.w) >> b.pixel_bits; };
    auto pixel_local_of = [&]() { return __float_as_uint(rd.w) & ((1u << b.pixel_bits) - 1u); };
    auto path_of = [&]() { return s_local_of() * b.n_pixels + pixel_local_of(); };  // slot in the radiance array
    float3 T = f3(th);
    if (h.y == REF_NONE) {  // src/camera.rs:347-349
        float3 c = T * cam.background;
        put4(&radiance[path_of()], make_float4(c.x, c.y, c.z, 0.0f));
        return false;
    }
    const uint4* meta = leaf_meta(sc, h.y);
    uint4 m0 = meta[0], m1 = meta[1];
    float3 o = f3(ro), d = f3(rd);
    HitGeom g = resolve_hit(sc, h.y, m0, o, d, ro.w, __uint_as_float(h.x));
    ShadeRec rec = shade_rec_of_leaf(sc, m0, m1);
    if (PLAIN_CHECKERS && ref_class(h.y) == CLS_CHECKER) {  // Texture::Checker over two plain colours (src/texture.rs:45-55)
        const DChecker ck = sc.checkers[m0.x];
        const bool is_even = checker_is_even(ck.scale, g.p);
        rec.colour = is_even ? f3(ck.even[0], ck.even[1], ck.even[2]) : f3(ck.odd[0], ck.odd[1], ck.odd[2]);
        rec.tex_kind = RT_TEX_SOLID;  // the texture has been evaluated
    }
    Rng rng;
    rng.init(path_key(seed, b.first_pixel + pixel_local_of(), b.first_sample + s_local_of()), bounce, STREAM_SCATTER);
    ScatterResult r = shade_hit<TEXTURES>(sc, rec, g, d, rng, lds_perlin);
    if (!r.scattered) {  // src/camera.rs:354-355
        float3 c = T * r.emitted;
        put4(&radiance[path_of()], make_float4(c.x, c.y, c.z, 0.0f));
        return false;
    }
    if (bounce + 1 >= cam.max_depth) {  // depth exhausted, src/camera.rs:342
        put4(&radiance[path_of()], make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        return false;
    }
    // throughput * attenuation + emitted (src/camera.rs:360), component by component
    nt = make_float4(T.x * r.attenuation.x + r.emitted.x, T.y * r.attenuation.y + r.emitted.y, T.z * r.attenuation.z + r.emitted.z, 0.0f);
    no = make_float4(g.p.x, g.p.y, g.p.z, ro.w);
    nd = make_float4(r.dir.x, r.dir.y, r.dir.z, rd.w);
    return true;
}

// Shade: Material::emitted / scatter for every (ray, hit) of the current queue.
//
// A block (512 threads) takes a tile of SHADE_TILE = 1024 consecutive queue slots, reads the leaf references of
// their hit records (the shading class of the leaf's material sits in bits of the reference) and counting-sorts
// the tile by class in LDS.  The threads then walk the tile in sorted order, so that the 64 lanes of a wave run
// the same material code on (almost always) the same class; the ray records they read all lie inside the tile,
// i.e. in cache lines the block consumes completely.  A tile's time is a chain of dependent memory round trips,
// so whatever can be fetched together is: the sorted references travel through LDS (ray, hit distance, the leaf's
// records, its primitive and placement then go out in one round trip), and the next tile's references are
// fetched while this tile's rays leave.  Hits on materials whose texture is an image, a noise or a checker of
// those are only listed here; k_shade_textured handles them, which keeps the register footprint of this kernel
// small (a checker of two plain colours is a class of its own and is shaded here).
// Measured on the headline scene (shade time per 16 samples): tiles of 256 / 512 / 1024 / 2048 slots take
// 36.7 / 23.8 / 22.9 / 26.6 ms - small tiles are bound by the queue atomic, large ones by LDS per block.
#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 512
#endif
#ifndef RT_SHADE_PER_THREAD
#define RT_SHADE_PER_THREAD 2
#endif
static constexpr int SHADE_BLOCK = RT_SHADE_BLOCK;
static constexpr int SHADE_PER_THREAD = RT_SHADE_PER_THREAD;
static constexpr int SHADE_TILE = SHADE_BLOCK * SHADE_PER_THREAD;

// Survivors of a tile are staged in LDS and appended to the next queue with ONE global atomic per
// tile (a per-wave atomic on the single queue counter serialises the whole chip: 2.5 M atomics
// per 4-spp pass of the 4K image cost more than all the shading), then copied out with
// consecutive lanes writing consecutive slots.
__global__ __launch_bounds__(SHADE_BLOCK) void k_shade(DScene sc, DCamera cam, BatchInfo b, uint64_t seed, uint32_t bounce,
                                                       uint32_t* __restrict__ counts, uint32_t* __restrict__ tex_count,
                                                       uint32_t* __restrict__ tex_list, const float4* __restrict__ ray_o,
                                                       const float4* __restrict__ ray_d, const float* __restrict__ thr,
                                                       const uint2* __restrict__ hits, float4* __restrict__ out_o,
                                                       float4* __restrict__ out_d, float* __restrict__ out_thr,
                                                       float4* __restrict__ radiance) {
    __shared__ float4 s_out[2][SHADE_TILE];        // new rays: origin + time, direction + path
    __shared__ float s_out_thr[3 * SHADE_TILE];    //           throughput
    __shared__ uint16_t s_order[SHADE_TILE];       // tile entries in class order
    __shared__ uint32_t s_ref[SHADE_TILE];         // and their leaf references
    __shared__ uint32_t s_count[N_SHADE_CLASSES];
    __shared__ uint32_t s_offset[N_SHADE_CLASSES + 1];
    __shared__ uint32_t s_alive, s_base, s_tex_base;
    const uint32_t n = counts[bounce];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    // leaf references of this thread's entries of the block's first tile (those of the next tile are fetched while the
    // current tile's rays leave: one memory round trip less in every tile's chain)
    uint32_t ref_in[SHADE_PER_THREAD];
#pragma unroll
    for (int k = 0; k < SHADE_PER_THREAD; ++k) {
        const uint32_t e0 = blockIdx.x * SHADE_TILE + k * SHADE_BLOCK + tid;
        ref_in[k] = e0 < n ? hits[e0].y : REF_NONE;
    }
    for (uint32_t tile = blockIdx.x * SHADE_TILE; tile < n; tile += gridDim.x * SHADE_TILE) {
        const uint32_t tile_n = min((uint32_t)SHADE_TILE, n - tile);
        if (tid < N_SHADE_CLASSES) s_count[tid] = 0;
        if (tid == 0) s_alive = 0;
        __syncthreads();
        // classify; rank inside (wave, class) by ballot, one LDS atomic per wave and class
        uint32_t cls[SHADE_PER_THREAD], pos[SHADE_PER_THREAD];
#pragma unroll
        for (int k = 0; k < SHADE_PER_THREAD; ++k) {
            uint32_t e = k * SHADE_BLOCK + tid;
            uint32_t c = N_SHADE_CLASSES;  // no entry
            if (e < tile_n) c = ref_in[k] == REF_NONE ? CLS_MISS : ref_class(ref_in[k]);
            cls[k] = c;
            pos[k] = 0;
            for (uint32_t q = 0; q < N_SHADE_CLASSES; ++q) {
                unsigned long long m = __ballot(c == q);
                if (m) {
                    uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u, base = 0;
                    if (lane == leader) base = atomicAdd(&s_count[q], (uint32_t)__popcll(m));
                    base = __shfl(base, (int)leader, 64);
                    if (c == q) pos[k] = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (uint32_t q = 0; q < N_SHADE_CLASSES; ++q) { s_offset[q] = run; run += s_count[q]; }
            s_offset[N_SHADE_CLASSES] = run;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SHADE_PER_THREAD; ++k)
            if (cls[k] < N_SHADE_CLASSES) {
                const uint32_t dst = s_offset[cls[k]] + pos[k];
                s_order[dst] = (uint16_t)(k * SHADE_BLOCK + tid);
                s_ref[dst] = ref_in[k];
            }
        __syncthreads();
        const uint32_t first_textured = s_offset[CLS_TEXTURED];

        for (int k = 0; k < SHADE_PER_THREAD; ++k) {
            const uint32_t e = k * SHADE_BLOCK + tid;
            bool alive = false;
            float4 no = make_float4(0, 0, 0, 0), nd = no, nt = no;
            if (e < tile_n) {
                const uint32_t i = tile + s_order[e];
                if (e < first_textured) {
                    // the leaf reference comes from LDS, so the ray, the hit distance, the leaf's records and its
                    // primitive are all fetched at once
                    const uint2 h = make_uint2(hits[i].x, s_ref[e]);
                    alive = shade_one<false>(sc, cam, b, seed, bounce, ray_o[i], ray_d[i], load_thr(thr, i), h, radiance, no, nd, nt);
                }
            }
            unsigned long long mask = __ballot(alive);
            if (mask) {
                uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u, slot0 = 0;
                if (lane == leader) slot0 = atomicAdd(&s_alive, (uint32_t)__popcll(mask));
                slot0 = __shfl(slot0, (int)leader, 64);
                if (alive) {
                    uint32_t slot = slot0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    put4(&s_out[0][slot], no);
                    put4(&s_out[1][slot], nd);
                    s_out_thr[3 * slot] = nt.x;
                    s_out_thr[3 * slot + 1] = nt.y;
                    s_out_thr[3 * slot + 2] = nt.z;
                }
            }
        }
        // the next tile's leaf references, on their way while this tile's rays are written out
        {
            const uint32_t next_tile = tile + gridDim.x * SHADE_TILE;
#pragma unroll
            for (int k = 0; k < SHADE_PER_THREAD; ++k) {
                const uint32_t e1 = next_tile + k * SHADE_BLOCK + tid;
                ref_in[k] = (next_tile < n && e1 < n) ? hits[e1].y : REF_NONE;
            }
        }
        __syncthreads();
        // one atomic per tile appends the new rays to the next queue, another (from another wave, at the same time)
        // the hits on textured materials - the last class of the sorted tile - to this bounce's list
        const uint32_t n_alive = s_alive, n_tex = tile_n - first_textured;
        if (tid == 0 && n_alive) s_base = atomicAdd(&counts[bounce + 1], n_alive);
        if (tid == 64 && n_tex) s_tex_base = atomicAdd(tex_count, n_tex);
        __syncthreads();
        for (uint32_t j = tid; j < n_tex; j += SHADE_BLOCK) tex_list[s_tex_base + j] = tile + s_order[first_textured + j];
        const uint32_t base = s_base;
        for (uint32_t j = tid; j < n_alive; j += SHADE_BLOCK) {
            out_o[base + j] = s_out[0][j];
            out_d[base + j] = s_out[1][j];
            store_thr(out_thr, base + j, make_float4(s_out_thr[3 * j], s_out_thr[3 * j + 1], s_out_thr[3 * j + 2], 0.0f));
        }
        __syncthreads();
    }
}

// The hits k_shade set aside: materials with checker, image or perlin-noise textures.  One global
// atomic per block appends the block's survivors.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_shade_textured(DScene sc, DCamera cam, BatchInfo b, uint64_t seed, uint32_t bounce,
                                                        uint32_t* __restrict__ counts, const uint32_t* __restrict__ tex_count,
                                                        const uint32_t* __restrict__ tex_list, const float4* __restrict__ ray_o,
                                                        const float4* __restrict__ ray_d, const float* __restrict__ thr,
                                                        const uint2* __restrict__ hits, float4* __restrict__ out_o,
                                                        float4* __restrict__ out_d, float* __restrict__ out_thr,
                                                        float4* __restrict__ radiance) {
    __shared__ uint32_t s_wave_count[4], s_base;
    __shared__ float4v s_vecs[LDS_PERLIN_VECS];
    __shared__ uint32_t s_perm[3 * 256];
    const uint32_t n = *tex_count;
    if (blockIdx.x * blockDim.x >= n) return;  // no work for this block (the whole block leaves)
    PerlinLds lds_perlin{(lds_const_float4v*)s_vecs, (lds_const_u32*)s_perm, false};
    if (sc.lds_perlin) {
        const DPerlin pn = sc.perlins[0];
        for (uint32_t i = threadIdx.x; i < pn.n_vecs; i += blockDim.x) {
            const float4 v = pn.vecs[i];
            s_vecs[i] = float4v{v.x, v.y, v.z, v.w};
        }
        for (uint32_t i = threadIdx.x; i < 3 * 256; i += blockDim.x) s_perm[i] = pn.perm[i];
        __syncthreads();
        lds_perlin.filled = true;
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t first = blockIdx.x * blockDim.x; first < n; first += gridDim.x * blockDim.x) {
        const uint32_t j = first + threadIdx.x;
        bool alive = false;
        float4 no = make_float4(0, 0, 0, 0), nd = no, nt = no;
        if (j < n) {
            const uint32_t i = tex_list[j];
            alive = shade_one<true>(sc, cam, b, seed, bounce, ray_o[i], ray_d[i], load_thr(thr, i), hits[i], radiance, no, nd, nt, lds_perlin);
        }
        unsigned long long mask = __ballot(alive);
        if (lane == 0) s_wave_count[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
            s_base = total ? atomicAdd(&counts[bounce + 1], total) : 0u;
        }
        __syncthreads();
        if (alive) {
            uint32_t slot = s_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            for (uint32_t w2 = 0; w2 < wave; ++w2) slot += s_wave_count[w2];
            put4(&out_o[slot], no);
            put4(&out_d[slot], nd);
            store_thr(out_thr, slot, nt);
        }
        __syncthreads();
    }
}

// Tail: every remaining bounce of the paths in a queue, in one launch.  A per-bounce pass costs tens of
// microseconds however short its queue is (launch, one chain of dependent node fetches, one tile's chain of
// barriers), and deep paths keep a few rays alive for dozens of bounces; once a batch is down to that regime the host
// stops launching per-bounce kernels and this kernel takes every path of the queue to its end: one thread per path,
// the loop of ray_colour (src/camera.rs:339-364) with the same device functions the per-bounce kernels use - the
// leaves kept out of the tree, the tree walk (stack in LDS), the shading of all material and texture classes - and
// the same random-number keys, so a path's radiance does not depend on where the hand-over happens.
static constexpr int TAIL_BLOCK = 256;
// Rays are dealt to the waves of the grid round-robin (ray i goes to wave i mod n_waves): a short queue is spread over
// the whole chip with a few rays per wave instead of filling a few waves.  A wave takes as long as its slowest lane in
// every part of a bounce, and runs the code of every material class its lanes have hit one after the other, so the time
// of a bounce grows with the number of lanes in use; the kernel's time is that of its longest path, bounce after bounce.
// A lane whose path has ended takes the wave's next ray, if there is one.
// GENERIC: as for k_extend; TEXTURES: the scene has hits of the textured class (image, noise, nested checkers).  The kernel
// waits for its own instruction stream more than for anything else, so a scene gets the variant with no more code than it needs.
template <bool GENERIC, bool TEXTURES>
__global__ __launch_bounds__(TAIL_BLOCK) void k_tail(DScene sc, DCamera cam, BatchInfo b, uint64_t seed, uint32_t first_bounce,
                                                     const uint32_t* __restrict__ counts, const float4* __restrict__ ray_o,
                                                     const float4* __restrict__ ray_d, const float* __restrict__ thr,
                                                     float4* __restrict__ radiance, unsigned long long* __restrict__ stack_overflow) {
    __shared__ unsigned long long s_stack[TAIL_BLOCK / 64][RT_EXTEND_STACK][64];
    const uint32_t n = counts[first_bounce];
    const uint32_t lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * (TAIL_BLOCK / 64) + wave_in_block, n_waves = gridDim.x * (TAIL_BLOCK / 64);
    if (wave >= n) return;
    WaveStack<RT_EXTEND_STACK> st;
    st.lds = (lds_u64*)&s_stack[wave_in_block][0][lane];
    st.ovf = (global_u64*)(stack_overflow + ((size_t)wave * OVERFLOW_STACK) * 64);
    st.lane = lane;
    Probe pr;
    unsigned long long next = (unsigned long long)wave + (unsigned long long)lane * n_waves;  // this lane's next ray
    const unsigned long long step = 64ull * n_waves;
    bool have = false;
    uint32_t bounce = first_bounce;
    float4 ro = make_float4(0, 0, 0, 0), rd = ro, th = ro;
#ifdef RT_SECTION_TIMES  // development aid: shader-clock cycles every 64th wave spends in each part of a bounce
    unsigned long long t_part[4] = {0, 0, 0, 0}, t_mark = __builtin_readcyclecounter();
    uint32_t rounds = 0, lane_rounds = 0;
    const unsigned long long t_begin = t_mark;
#define RT_TAIL_PART_END(i) { const unsigned long long t_now = __builtin_readcyclecounter(); t_part[i] += t_now - t_mark; t_mark = t_now; }
#else
#define RT_TAIL_PART_END(i)
#endif
    for (;;) {
        if (!have && next < n) {
            const uint32_t i = (uint32_t)next;
            ro = ray_o[i]; rd = ray_d[i]; th = load_thr(thr, i);
            bounce = first_bounce;
            have = true;
            next += step;
        }
        if (__ballot(have) == 0ull) break;  // no lane has a ray and none could fetch one
#ifdef RT_SECTION_TIMES
        rounds++; lane_rounds += (uint32_t)__popcll(__ballot(have));
#endif
        RT_TAIL_PART_END(0)
#ifdef RT_SECTION_TIMES  // (the same bounce with the clock read where all lanes meet again)
        {
            BatchMediumKey key{seed, __float_as_uint(rd.w), bounce, b};
            TravRay r = make_trav_ray(f3(ro), f3(rd), ro.w, RENDER_TMIN);
            float tmax = RT_INF;
            uint32_t best = REF_NONE;
            if (have) global_leaves<false, GENERIC>(sc, r, tmax, best, key, pr);
            RT_TAIL_PART_END(1)
            if (have) {
                st.sp = 0;
                walk_tree<false, GENERIC>(sc, r, tmax, best, key, st, pr);
            }
            RT_TAIL_PART_END(2)
            if (have) {
                float4 no = make_float4(0, 0, 0, 0), nd = no, nt = no;
                have = shade_one<TEXTURES, true>(sc, cam, b, seed, bounce, ro, rd, th, make_uint2(__float_as_uint(tmax), best), radiance, no, nd, nt);
                ro = no; rd = nd; th = nt;
                ++bounce;
            }
            RT_TAIL_PART_END(3)
        }
#else
        if (have) {
            const BatchMediumKey key{seed, __float_as_uint(rd.w), bounce, b};
            const TravRay r = make_trav_ray(f3(ro), f3(rd), ro.w, RENDER_TMIN);
            float tmax = RT_INF;
            uint32_t best = REF_NONE;
            global_leaves<false, GENERIC>(sc, r, tmax, best, key, pr);
            st.sp = 0;
            walk_tree<false, GENERIC>(sc, r, tmax, best, key, st, pr);
            float4 no = make_float4(0, 0, 0, 0), nd = no, nt = no;
            // (shade_one ends the path - and writes its radiance - at a miss, a light, an absorbed ray or the depth limit)
            have = shade_one<TEXTURES, true>(sc, cam, b, seed, bounce, ro, rd, th, make_uint2(__float_as_uint(tmax), best), radiance, no, nd, nt);
            ro = no; rd = nd; th = nt;
            ++bounce;
        }
#endif
    }
#ifdef RT_SECTION_TIMES
    if (lane == 0 && (wave & 63u) == 0)
        printf("tailparts %u total %llu fetch %llu leaves %llu walk %llu shade %llu rounds %u lanes %u\n", wave, __builtin_readcyclecounter() - t_begin, t_part[0], t_part[1],
               t_part[2], t_part[3], rounds, lane_rounds);
#endif
}

// acc[pixel] += colour, one sample after the other (src/camera.rs:295-296)
__global__ __launch_bounds__(256) void k_accumulate(BatchInfo b, const float4* __restrict__ radiance, float4* __restrict__ acc) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < b.n_pixels; p += gridDim.x * blockDim.x) {
        float4 a = acc[b.first_pixel + p];
        for (uint32_t s = 0; s < b.n_samples; ++s) {
            float4 c = radiance[(size_t)s * b.n_pixels + p];
            a.x += c.x;
            a.y += c.y;
            a.z += c.z;
        }
        acc[b.first_pixel + p] = a;
    }
}

// acc[pixel] += part[pixel] (x, y, z; lane 3 is left alone): the sum step of a render that was split by sample number
// over several GPUs, applied in GPU order.
__global__ __launch_bounds__(256) void k_add_buffers(float4* __restrict__ acc, const float4* __restrict__ part, unsigned long long n_pixels) {
    for (unsigned long long p = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; p < n_pixels;
         p += (unsigned long long)gridDim.x * blockDim.x) {
        float4 a = acc[p];
        const float4 b = part[p];
        a.x += b.x; a.y += b.y; a.z += b.z;
        acc[p] = a;
    }
}

// Colour::to_rgba (src/math/vec3.rs:252-259)
RT_D uint32_t to_rgba(float3 c) {
    float r = sqrtf(c.x) * 255.999f, g = sqrtf(c.y) * 255.999f, bl = sqrtf(c.z) * 255.999f;
    uint32_t ri = rust_as_u32(r), gi = rust_as_u32(g), bi = rust_as_u32(bl);
    return (ri << 0) | (gi << 8) | (bi << 16);
}

__global__ __launch_bounds__(256) void k_finalize(const float4* __restrict__ acc, unsigned long long n_pixels, float inv_samples, int mode,
                                                  float exposure, void* __restrict__ out) {
    for (unsigned long long p = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; p < n_pixels;
         p += (unsigned long long)gridDim.x * blockDim.x) {
        float4 a = acc[p];
        float3 c = inv_samples * f3(a);
        if (mode == 0) {
            ((float4*)out)[p] = make_float4(c.x, c.y, c.z, 0.0f);
        } else if (mode == 1) {
            ((float4*)out)[p] = make_float4(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z), 0.0f);
        } else {
            float3 m = f3(1.0f - expf(exposure * -c.x), 1.0f - expf(exposure * -c.y), 1.0f - expf(exposure * -c.z));
            ((uint32_t*)out)[p] = to_rgba(m);
        }
    }
}

// Ray::hit_anything on caller-supplied rays; fills complete hit records.
__global__ __launch_bounds__(256) void k_intersect(DScene sc, uint64_t seed, const rt_ray* __restrict__ rays, unsigned long long n,
                                                   rt_hit* __restrict__ out) {
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        rt_ray r = rays[i];
        float3 o = f3(r.origin[0], r.origin[1], r.origin[2]), d = f3(r.direction[0], r.direction[1], r.direction[2]);
        MediumKey key{seed, r.rng_pixel, r.rng_sample, r.rng_bounce};
        Probe pr;
        float t;
        uint32_t ref = closest_hit<false>(sc, o, d, r.time, r.tmin, r.tmax, key, t, pr);
        rt_hit h = rt_hit{};
        if (ref != REF_NONE) {
            HitGeom g = resolve_hit(sc, ref, leaf_meta(sc, ref)[0], o, d, r.time, t);
            ensure_uv(g);
            h.t = t;
            h.point[0] = g.p.x; h.point[1] = g.p.y; h.point[2] = g.p.z;
            h.normal[0] = g.normal.x; h.normal[1] = g.normal.y; h.normal[2] = g.normal.z;
            h.u = g.u; h.v = g.v;
            h.prim_id = g.prim_id;
            h.material = g.material;
            h.front_face = g.front_face ? 1u : 0u;
            h.hit = 1u;
        }
        out[i] = h;
    }
}

// The render path's traversal on caller-supplied rays (per-ray parity checks of k_extend): rays enter the queue
// layout here, k_first_leaves / k_extend run on the queue as they do in a render, and the hit records are
// completed afterwards.  A ray's path id is (rng_sample << pixel_bits) | rng_pixel, so the medium draws get the
// (pixel, sample) key the caller gave.
__global__ __launch_bounds__(256) void k_pack_rays(const rt_ray* __restrict__ rays, unsigned long long n, uint32_t pixel_bits,
                                                   float4* __restrict__ ray_o, float4* __restrict__ ray_d) {
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const rt_ray r = rays[i];
        const uint32_t id = (r.rng_sample << pixel_bits) | r.rng_pixel;
        ray_o[i] = make_float4(r.origin[0], r.origin[1], r.origin[2], r.time);
        ray_d[i] = make_float4(r.direction[0], r.direction[1], r.direction[2], __uint_as_float(id));
    }
}

RT_D rt_hit hit_record_of(const DScene& sc, uint32_t ref, float3 o, float3 d, float time, float t) {
    rt_hit h = rt_hit{};
    if (ref != REF_NONE) {
        HitGeom g = resolve_hit(sc, ref, leaf_meta(sc, ref)[0], o, d, time, t);
        ensure_uv(g);
        h.t = t;
        h.point[0] = g.p.x; h.point[1] = g.p.y; h.point[2] = g.p.z;
        h.normal[0] = g.normal.x; h.normal[1] = g.normal.y; h.normal[2] = g.normal.z;
        h.u = g.u; h.v = g.v;
        h.prim_id = g.prim_id;
        h.material = g.material;
        h.front_face = g.front_face ? 1u : 0u;
        h.hit = 1u;
    }
    return h;
}

__global__ __launch_bounds__(256) void k_unpack_hits(DScene sc, const rt_ray* __restrict__ rays, const uint2* __restrict__ hits,
                                                     unsigned long long n, rt_hit* __restrict__ out) {
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const rt_ray r = rays[i];
        const uint2 h = hits[i];
        out[i] = hit_record_of(sc, h.y, f3(r.origin[0], r.origin[1], r.origin[2]), f3(r.direction[0], r.direction[1], r.direction[2]), r.time,
                               __uint_as_float(h.x));
    }
}

// Material::emitted/scatter on caller-supplied (ray, hit) pairs: per-material parity checks.
__global__ __launch_bounds__(256) void k_shade_records(DScene sc, uint64_t seed, const rt_ray* __restrict__ rays,
                                                       const rt_hit* __restrict__ hits, unsigned long long n,
                                                       rt_ray* __restrict__ out_rays, float* __restrict__ out_att,
                                                       float* __restrict__ out_emit) {
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        rt_ray r = rays[i];
        rt_hit h = hits[i];
        HitGeom g;
        g.p = f3(h.point[0], h.point[1], h.point[2]);
        g.normal = f3(h.normal[0], h.normal[1], h.normal[2]);
        g.u = h.u; g.v = h.v;
        g.uv_ready = true;
        g.uv_dir = f3(0, 0, 0);
        g.material = h.material;
        g.prim_id = h.prim_id;
        g.front_face = h.front_face != 0u;
        Rng rng;
        rng.init(path_key(seed, r.rng_pixel, r.rng_sample), r.rng_bounce, STREAM_SCATTER);
        ShadeRec rec = shade_rec_of_material(sc, h.material);
        ScatterResult s = shade_hit<true>(sc, rec, g, f3(r.direction[0], r.direction[1], r.direction[2]), rng);
        rt_ray o = rt_ray{};
        if (s.scattered) {
            o.origin[0] = g.p.x; o.origin[1] = g.p.y; o.origin[2] = g.p.z; o.time = r.time;
            o.direction[0] = s.dir.x; o.direction[1] = s.dir.y; o.direction[2] = s.dir.z; o.tmin = 1.0f;
        } else {
            s.attenuation = f3(0, 0, 0);
        }
        out_rays[i] = o;
        out_att[3 * i] = s.attenuation.x; out_att[3 * i + 1] = s.attenuation.y; out_att[3 * i + 2] = s.attenuation.z;
        out_emit[3 * i] = s.emitted.x; out_emit[3 * i + 1] = s.emitted.y; out_emit[3 * i + 2] = s.emitted.z;
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline uint32_t grid_for(unsigned long long n, uint32_t max_blocks) {
    unsigned long long blocks = (n + 255ull) / 256ull;
    if (blocks < 1) blocks = 1;
    if (blocks > max_blocks) blocks = max_blocks;
    return (uint32_t)blocks;
}

void launch_raygen(hipStream_t st, uint32_t max_blocks, const DScene& sc, const DCamera& cam, const BatchInfo& b, uint64_t seed, float4* ray_o,
                   float4* ray_d, float* thr, uint2* first_hits, uint32_t* counts, uint32_t n_counts) {
    uint32_t total = b.n_pixels * b.n_samples;
    hipLaunchKernelGGL(sc.generic_media ? k_raygen<true> : k_raygen<false>, dim3(grid_for(total, max_blocks)), dim3(256), 0, st, sc, cam, b, seed, ray_o, ray_d, thr,
                       sc.n_global != 0 ? first_hits : nullptr, counts, n_counts);
}

void launch_primary_rays(hipStream_t st, uint32_t max_blocks, const DCamera& cam, uint64_t seed, uint32_t sample, const uint32_t* pixels,
                         unsigned long long n, rt_ray* out) {
    hipLaunchKernelGGL(k_primary_rays, dim3(grid_for(n, max_blocks)), dim3(256), 0, st, cam, seed, sample, pixels, n, out);
}

// Blocks of the extend kernel (plain or counting build) that one CU holds at a time, as the runtime computes it from the
// kernel's registers, LDS and scratch.  The persistent grid must not exceed CUs x this number:
// blocks beyond it only start when earlier ones have finished, and then work their interleaved
// share of the queue on an almost empty chip.
int extend_blocks_per_cu(bool counting, bool generic_media) {
    int a = 0;
    // the smaller figure of the two variants (large / small tree) a scene of this kind can get
    auto blocks_of = [](auto kernel) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, EXTEND_BLOCK, 0) != hipSuccess) n = 0;
        return n;
    };
    if (generic_media)
        a = counting ? std::min(blocks_of(k_extend<true, false, true>), blocks_of(k_extend<true, true, true>))
                     : std::min(blocks_of(k_extend<false, false, true>), blocks_of(k_extend<false, true, true>));
    else
        a = counting ? std::min(blocks_of(k_extend<true, false, false>), blocks_of(k_extend<true, true, false>))
                     : std::min(blocks_of(k_extend<false, false, false>), blocks_of(k_extend<false, true, false>));
    (void)hipGetLastError();
    // a block whose LDS divides the CU's 160 KiB exactly does not get the last slot in practice
    int by_lds = (int)((160u * 1024u - 1024u) / (512u * (EXTEND_BLOCK / 64) * (uint32_t)RT_EXTEND_STACK + 32u));  // 512 B per stack level and wave
    return std::min(a, by_lds);
}
int tail_blocks_per_cu() {
    int a = 0;
    a = 1 << 30;
    for (auto kernel : {k_tail<false, false>, k_tail<false, true>, k_tail<true, false>, k_tail<true, true>}) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, TAIL_BLOCK, 0) != hipSuccess) n = 0;
        a = std::min(a, n);
    }
    (void)hipGetLastError();
    // no more than three blocks per CU even where four fit: measured on the Book-1 scene at 400x225, depth 50, 100 samples per call
    // (3.49 ms per call with three, 3.68 with four, 3.88 with two); the interactive frames differ by 2 - 4 % either way
    return std::min(a, 3);
}
int extend_lds_stack_levels(uint32_t n_nodes) {
    return n_nodes <= (uint32_t)SMALL_TREE_NODES ? ExtendLds<true>::STACK_LEVELS : ExtendLds<false>::STACK_LEVELS;
}
// bytes of the extend kernel's stack overflow area for a grid of `blocks` blocks
size_t extend_overflow_bytes(uint32_t blocks) { return (size_t)blocks * (EXTEND_BLOCK / 64) * OVERFLOW_STACK * 64 * sizeof(unsigned long long); }

void launch_first_leaves(hipStream_t st, uint32_t max_blocks, uint32_t upper_bound, const DScene& sc, const BatchInfo& b, uint64_t seed,
                         uint32_t bounce, const uint32_t* counts, const float4* ray_o, const float4* ray_d, uint2* hits, DCounters* counters) {
    dim3 g0(grid_for(upper_bound, max_blocks * 4u));
    auto kernel = counters ? (sc.generic_media ? k_first_leaves<true, true> : k_first_leaves<true, false>)
                           : (sc.generic_media ? k_first_leaves<false, true> : k_first_leaves<false, false>);
    hipLaunchKernelGGL(kernel, g0, dim3(256), 0, st, sc, b, seed, bounce, counts, ray_o, ray_d, hits, counters);
}

void launch_extend(hipStream_t st, uint32_t resident_blocks, uint32_t upper_bound, const DScene& sc, const BatchInfo& b, uint64_t seed,
                   uint32_t bounce, const uint32_t* counts, const float4* ray_o, const float4* ray_d, uint2* hits,
                   unsigned long long* stack_overflow, const ExtendTuning& tune, DCounters* counters) {
    dim3 grid(grid_for(upper_bound, resident_blocks));
    const bool small_tree = sc.n_nodes <= (uint32_t)SMALL_TREE_NODES;
    // stack_overflow == nullptr: the caller's word that no walk of this tree stacks more entries than the kernel keeps in LDS
    auto kernel = sc.generic_media
        ? (counters ? (small_tree ? k_extend<true, true, true> : k_extend<true, false, true>) : (small_tree ? k_extend<false, true, true> : k_extend<false, false, true>))
        : (counters ? (small_tree ? k_extend<true, true, false> : k_extend<true, false, false>) : (small_tree ? k_extend<false, true, false> : k_extend<false, false, false>));
    if (!counters && stack_overflow == nullptr)
        kernel = sc.generic_media ? (small_tree ? k_extend<false, true, true, false> : k_extend<false, false, true, false>)
                                  : (small_tree ? k_extend<false, true, false, false> : k_extend<false, false, false, false>);
    hipLaunchKernelGGL(kernel, grid, dim3(EXTEND_BLOCK), 0, st, sc, b, seed, bounce, counts, ray_o, ray_d, hits, stack_overflow, tune, counters);
}

void launch_shade(hipStream_t st, uint32_t max_blocks, uint32_t upper_bound, const DScene& sc, const DCamera& cam, const BatchInfo& b,
                  uint64_t seed, uint32_t bounce, uint32_t* counts, uint32_t* tex_count, uint32_t* tex_list, const float4* ray_o,
                  const float4* ray_d, const float* thr, const uint2* hits, float4* out_o, float4* out_d, float* out_thr, float4* radiance,
                  bool scene_has_textures) {
    unsigned long long tiles = ((unsigned long long)upper_bound + SHADE_TILE - 1) / SHADE_TILE;
    uint32_t grid = (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>(tiles, 1), max_blocks);
    hipLaunchKernelGGL(k_shade, dim3(grid), dim3(SHADE_BLOCK), 0, st, sc, cam, b, seed, bounce, counts, tex_count, tex_list, ray_o, ray_d, thr,
                       hits, out_o, out_d, out_thr, radiance);
    if (scene_has_textures)
        hipLaunchKernelGGL(k_shade_textured, dim3(grid_for(upper_bound, max_blocks)), dim3(256), 0, st, sc, cam, b, seed, bounce, counts,
                           tex_count, tex_list, ray_o, ray_d, thr, hits, out_o, out_d, out_thr, radiance);
}

void launch_tail(hipStream_t st, uint32_t resident_blocks, uint32_t upper_bound, const DScene& sc, const DCamera& cam, const BatchInfo& b,
                 uint64_t seed, uint32_t first_bounce, const uint32_t* counts, const float4* ray_o, const float4* ray_d, const float* thr,
                 float4* radiance, unsigned long long* stack_overflow) {
    // as many waves as the chip holds, unless there are fewer rays than that (the fewer lanes of a wave have a ray, the
    // shorter its bounce)
    const uint32_t blocks = (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>(((unsigned long long)upper_bound + 3ull) / 4ull, 1ull), resident_blocks);
    auto kernel = sc.generic_media ? (sc.textured ? k_tail<true, true> : k_tail<true, false>) : (sc.textured ? k_tail<false, true> : k_tail<false, false>);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(TAIL_BLOCK), 0, st, sc, cam, b, seed, first_bounce, counts, ray_o,
                       ray_d, thr, radiance, stack_overflow);
}

void launch_accumulate(hipStream_t st, uint32_t max_blocks, const BatchInfo& b, const float4* radiance, float4* acc) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(b.n_pixels, max_blocks)), dim3(256), 0, st, b, radiance, acc);
}

void launch_add_buffers(hipStream_t st, uint32_t max_blocks, float4* acc, const float4* part, unsigned long long n_pixels) {
    hipLaunchKernelGGL(k_add_buffers, dim3(grid_for(n_pixels, max_blocks)), dim3(256), 0, st, acc, part, n_pixels);
}

void launch_finalize(hipStream_t st, uint32_t max_blocks, const float4* acc, unsigned long long n_pixels, float inv_samples, int mode,
                     float exposure, void* out) {
    hipLaunchKernelGGL(k_finalize, dim3(grid_for(n_pixels, max_blocks)), dim3(256), 0, st, acc, n_pixels, inv_samples, mode, exposure, out);
}

void launch_intersect(hipStream_t st, uint32_t max_blocks, const DScene& sc, uint64_t seed, const rt_ray* rays, unsigned long long n,
                      rt_hit* out) {
    hipLaunchKernelGGL(k_intersect, dim3(grid_for(n, max_blocks)), dim3(256), 0, st, sc, seed, rays, n, out);
}

void launch_pack_rays(hipStream_t st, uint32_t max_blocks, const rt_ray* rays, unsigned long long n, uint32_t pixel_bits, float4* ray_o,
                      float4* ray_d) {
    hipLaunchKernelGGL(k_pack_rays, dim3(grid_for(n, max_blocks)), dim3(256), 0, st, rays, n, pixel_bits, ray_o, ray_d);
}

void launch_unpack_hits(hipStream_t st, uint32_t max_blocks, const DScene& sc, const rt_ray* rays, const uint2* hits, unsigned long long n,
                        rt_hit* out) {
    hipLaunchKernelGGL(k_unpack_hits, dim3(grid_for(n, max_blocks)), dim3(256), 0, st, sc, rays, hits, n, out);
}

void launch_shade_records(hipStream_t st, uint32_t max_blocks, const DScene& sc, uint64_t seed, const rt_ray* rays,
                          const rt_hit* hits, unsigned long long n, rt_ray* out_rays, float* out_att, float* out_emit) {
    hipLaunchKernelGGL(k_shade_records, dim3(grid_for(n, max_blocks)), dim3(256), 0, st, sc, seed, rays, hits, n, out_rays, out_att, out_emit);
}

}  // namespace rt
