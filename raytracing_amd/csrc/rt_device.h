// rt_device.h — device-side data layout and math/RNG helpers of the gfx950
// path-tracing backend.  Included by the kernels (device code) and by the host
// code that fills the buffers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_HD __host__ __device__ __forceinline__
#define RT_D __device__ __forceinline__

namespace rt {

// ---------------------------------------------------------------------------
// Leaf / child references.  bit 31 set: leaf, bits 29..30 leaf kind (the
// rt_prim_kind values), bit 28 "placed" (the leaf is stored in object space and
// carries a Move/RotateY placement), bits 25..27 the shading class of the leaf's material
// (CLS_* below: shade sorts hits by it straight from the hit record), bits 0..24 index into
// that kind's array.  bit 31 clear: index of an inner BVH node.  REF_NONE: empty slot.
// ---------------------------------------------------------------------------
static constexpr uint32_t REF_LEAF = 0x80000000u;
static constexpr uint32_t REF_PLACED = 0x10000000u;
static constexpr uint32_t REF_NONE = 0xFFFFFFFFu;
static constexpr uint32_t REF_CLASS_SHIFT = 25;
static constexpr uint32_t MAX_PRIMS_PER_KIND = 1u << REF_CLASS_SHIFT;
static constexpr uint32_t KIND_SPHERE = 0, KIND_MOVING = 1, KIND_QUAD = 2, KIND_MEDIUM = 3;
RT_HD uint32_t make_leaf(uint32_t kind, uint32_t index, bool placed, uint32_t shading_class = 0) {
    return REF_LEAF | (kind << 29) | (placed ? REF_PLACED : 0u) | (shading_class << REF_CLASS_SHIFT) | index;
}
RT_HD bool ref_is_leaf(uint32_t r) { return (r & REF_LEAF) != 0; }
RT_HD uint32_t ref_kind(uint32_t r) { return (r >> 29) & 3u; }
RT_HD uint32_t ref_class(uint32_t r) { return (r >> REF_CLASS_SHIFT) & 7u; }
RT_HD uint32_t ref_index(uint32_t r) { return r & (MAX_PRIMS_PER_KIND - 1u); }
RT_HD bool ref_placed(uint32_t r) { return (r & REF_PLACED) != 0; }

// ---------------------------------------------------------------------------
// HBM layout (all arrays 16-byte aligned, one or more float4 per element):
//   nodes   4 x float4 per node of four children (64 B, rt_bvh.h PackedNode): the node's own box as an origin and a
//           step per axis, and the children's boxes as 8-bit coordinates on that grid, by axis and side (four
//           children to a 32-bit word), so that a ray picks the words of the sides it meets first / last by the sign
//           of its direction:
//             [0] = lo.x, lo.y, lo.z, step.x      [1] = step.y, step.z, bytes(lower x), bytes(upper x)
//             [2] = bytes(lower y), bytes(upper y), bytes(lower z), bytes(upper z)      [3] = child references
//           A child's grid box contains the child's box with a margin; the node test is conservative (it decides
//           what is looked at, never what is hit).  A leaf whose box is the reference's own AABB has that box in
//           *_boxes, and a hit on it only counts if the reference's box test of that box passes - the test that
//           decides in the reference's tree whether the leaf is looked at.  An empty slot repeats the first child's
//           reference under a box that is inside out.  (The builder's array, rt_bvh.h, keeps f32 boxes by child;
//           rt_scene_create packs it.)  With RT_BVH_WIDTH=2 a node is 4 x float4 of f32 boxes:
//             [0] = l.min.xy, l.max.xy   [1] = r.min.xy, r.max.xy
//             [2] = l.min.z, l.max.z, r.min.z, r.max.z   [3] = bits(left), bits(right), slacks
//           (slack multiplies the child's exit distance: 1 for a leaf whose box is the reference's own AABB, 1 + 2 ulp
//           for conservative boxes.)  Nodes are numbered breadth-first for the first levels (the part of the tree
//           every ray walks sits in a few KB at the start of the array), depth-first below.
//   *_boxes 2 x float4 per sphere, moving sphere and quad: min.xyz, max.xyz of the leaf's own AABB as the reference
//           computes it (Hittable::calc_aabb, src/hittable.rs:158-178); meaningful for leaves that are not placed
//   spheres 1 x float4: centre.xyz, radius                      (16 B)
//   moving  2 x float4: centre0.xyz, radius | dcentre.xyz, 0     (32 B)
//   quads   4 x float4: N.xyz, D | q.xyz, w.x | u.xyz, w.y | v.xyz, w.z  (64 B)
//           N = unit(u x v), D = N.q, w = n/(n.n): the per-call constants of Quad::hit
//           (src/hittable.rs:373-375); alpha = w.((p-q) x v), beta = w.(u x (p-q)) are then
//           evaluated as the reference evaluates them (:389-391)
//   *_meta  2 x uint4 per leaf (spheres, moving spheres, quads, media): what shading needs in
//           one 32-byte read: material, prim_id, transform (or RT_NO_TRANSFORM),
//           class | material kind << 8 | texture kind << 16  ||  bits(param), bits(r, g, b of a
//           solid-colour texture)
//   xforms  2 x float4 per placement: sin, cos, 0, 0 | offset.xyz, 0; *_xf: the same per leaf
// A leaf wrapped in Move/RotateY stays in object space; the ray is taken into
// object space at the leaf exactly as Ray::hit_anything does (src/math/ray.rs:207-247),
// which keeps hit distances bit-identical to the reference's.  Its BVH box is
// the world-space box of the placed leaf.
// ---------------------------------------------------------------------------
static constexpr uint32_t XFORM_MIXED = 0xFFFFFFFEu;  // DMedium::shared_xform: the boundary leaves do not share one placement
struct DMedium {
    uint32_t first_boundary, n_boundary;
    float neg_inv_density;
    uint32_t phase_material;
    uint32_t prim_id;
    uint32_t single_sphere;  // 1: the boundary is one sphere in world space, repeated below (saves two dependent reads)
    uint32_t shared_xform;   // the placement all boundary leaves share (RT_NO_TRANSFORM: none is placed), or XFORM_MIXED
    uint32_t _pad;
    float4 sphere;           // centre.xyz, radius of that sphere
};

struct DMaterial {
    uint32_t kind;
    float param;
    uint32_t texture;
    uint32_t _pad;
};

struct DTexture {
    uint32_t kind;
    float r, g, b;
    float scale;
    uint32_t even, odd;
    uint32_t image, perlin;
    uint32_t _pad[3];
};

// Texture::Checker (src/texture.rs:45-55) whose two sub-textures are plain colours, by material index.
struct DChecker {
    float scale;
    float even[3], odd[3];
    float _pad;
};

struct DImage {
    const void* texels;  // uchar4 (RGBX) or float4 per texel
    uint32_t width, height;
    uint32_t format;     // rt_image_format
    uint32_t _pad;
};

// PerlinNoise's tables (src/perlin_noise.rs:17-37), laid out so that a lookup needs no remainder operation:
// the reference indexes perm_x[(i & 255) % n] and rand_vecs[(px ^ py ^ pz) % n] with n = point_count.
struct DPerlin {
    const float4* vecs;      // vecs[v] = rand_vecs[v % n] (xyz, 0) for v below the power of two >= n: the xor of three
                             // permutation entries (each < n) stays below that power of two
    const uint32_t* perm;    // 3 * 256: perm[256 a + v] = perm_a[v % n] for axis a = x, y, z and v = 0..255
    uint32_t point_count;
    uint32_t n_vecs;         // entries of vecs: the power of two >= point_count
};
static constexpr uint32_t LDS_PERLIN_VECS = 256;  // largest n_vecs the LDS copy holds (the reference's tables have 64 - 256 points)

struct DScene {
    const float4* nodes;
    const float4* spheres;
    const uint4* sphere_meta;
    const float4* moving;
    const uint4* moving_meta;
    const float4* quads;
    const uint4* quad_meta;
    const DMedium* media;
    const uint4* medium_meta;
    const uint32_t* boundary_refs;
    const float4* boundary_boxes;  // 2 x float4 per boundary reference: min, max of the leaf's own box in the leaf's space
    const float4* xforms;      // 2 x float4 per placement
    // the placement of leaf i once more at [2 i], [2 i + 1] (NULL when no leaf of the kind is placed):
    // a leaf test reads it together with the primitive instead of after it
    const float4* sphere_xf;
    const float4* moving_xf;
    const float4* quad_xf;
    // the reference's own AABB of leaf i at [2 i] (min), [2 i + 1] (max)
    const float4* sphere_boxes;
    const float4* moving_boxes;
    const float4* quad_boxes;
    const DMaterial* materials;
    const DTexture* textures;
    const DImage* images;
    const DPerlin* perlins;
    const DChecker* checkers;  // n_materials entries, meaningful for the materials of CLS_CHECKER leaves
    uint32_t root;             // child reference of the tree (REF_NONE when it is empty)
    uint32_t n_nodes;
    uint32_t n_media;
    uint32_t n_materials;
    uint32_t n_global;         // leaves kept out of the tree: tested first for every ray
    uint32_t global_inside;    // bit g: the tree's leaves all lie inside the box of global_refs[g] (a sphere): rays start inside that box
    uint32_t lds_perlin;       // 1: the scene has exactly one noise table and it fits the textured-shade kernel's LDS copy
    uint32_t textured;         // 1: some leaf's hits are of the textured shading class
    uint32_t generic_media;    // 1: some medium's boundary is not one sphere in world space (selects the kernels that hold the code for those)
    uint32_t plain_sphere_boxes;  // 1: every sphere's own box is (centre - |radius|, centre + |radius|) as f32 arithmetic gives it (no box
                                  // was padded): the kernels compute it from the sphere's record and sphere_boxes is NULL
    const uint32_t* global_refs;  // n_global leaf references tested ahead of the tree
};

// Shading classes: rays of one class run the same code, so the shade kernel groups the rays of
// a tile by class before it processes them.
static constexpr uint32_t CLS_MISS = 0, CLS_LIGHT = 1, CLS_LAMBERTIAN = 2, CLS_METAL = 3, CLS_DIELECTRIC = 4, CLS_ISOTROPIC = 5,
                          CLS_CHECKER = 6,   // lambertian whose texture is a checker of two plain colours (DChecker): shaded inline
                          CLS_TEXTURED = 7,  // every other texture that is not a plain colour: set aside for k_shade_textured (last class)
                          N_SHADE_CLASSES = 8;

struct DCamera {
    float3 centre, pixel00, du, dv, ddu, ddv, background;
    float defocus_angle;
    uint32_t width, height, max_depth;
};

struct DCounters {
    unsigned long long samples, segments, node_visits, sphere_tests, quad_tests, medium_tests, hits;
    unsigned long long max_stack;  // deepest traversal stack any ray of the extend kernel has had (entries)
};

// ---------------------------------------------------------------------------
// float3 helpers (HIP's float3 already has the component-wise operators)
// ---------------------------------------------------------------------------
RT_HD float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
RT_HD float3 f3(float4 v) { return make_float3(v.x, v.y, v.z); }
RT_HD float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD float3 cross3(float3 l, float3 r) {
    return f3(l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x);
}
RT_HD float len3(float3 a) { return sqrtf(dot3(a, a)); }
RT_HD float3 unit3(float3 a) { return (1.0f / len3(a)) * a; }

// ---------------------------------------------------------------------------
// Counter-based random numbers: a pure function of
// (seed, pixel, sample, bounce, stream); draws inside one (bounce, stream) are
// sequential xoroshiro64**.  Replaces the per-row xoshiro256++ stream of
// src/rng.rs:17-60 / src/camera.rs:287 so that results do not depend on the
// order in which rays are processed.
// ---------------------------------------------------------------------------
static constexpr uint32_t STREAM_CAMERA = 0, STREAM_SCATTER = 1, STREAM_MEDIUM0 = 2;

RT_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
RT_HD uint64_t path_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    return mix64(seed + 0x9E3779B97F4A7C15ull * (((uint64_t)pixel << 32) | sample));
}
struct Rng {
    uint32_t s0, s1;
    RT_HD void init(uint64_t pkey, uint32_t bounce, uint32_t stream) {
        uint64_t b = mix64(pkey ^ (0xD1B54A32D192ED03ull * ((((uint64_t)bounce << 32) | stream) + 1ull)));
        s0 = (uint32_t)b;
        s1 = (uint32_t)(b >> 32);
        if ((s0 | s1) == 0u) s0 = 0x9E3779B9u;
    }
    RT_HD static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    RT_HD uint32_t next_u32() {
        uint32_t r = rotl(s0 * 0x9E3779BBu, 5) * 5u;
        s1 ^= s0;
        s0 = rotl(s0, 26) ^ s1 ^ (s1 << 9);
        s1 = rotl(s1, 13);
        return r;
    }
    RT_HD float next() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
};

}  // namespace rt
