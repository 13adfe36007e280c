// oracle_core.hpp — CPU restatement of rookieCookies/raytracing's render path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the shipped GPU path may include, link
// or call this code; it exists so that tests and the benchmark's CPU-baseline
// leg have something to check the HIP kernels against.
//
// PARITY UNPINNED BY THE REFERENCE: the reference (a nightly-only Rust crate
// with un-vendored git/crates.io dependencies) cannot be built in this image
// and ships no test, golden vector or fixture for its render path (its only
// tests cover an unused Matrix type, src/math/matrix.rs:111-219).  This file
// therefore restates the reference's algorithm function by function, each
// with the reference file:line it follows, in scalar f32 with the reference's
// operation order (compile with -ffp-contract=off: rustc does not contract
// a*b+c).  The pins that exist are this restatement's own known answers
// (tests/golden) and closed-form cases.
//
// Two random-number modes:
//   RngXoshiro — the reference's stream: one xoshiro256++ state per image
//                row, seeded [y, n_samples, W*H, y] (src/camera.rs:287),
//                consumed in program order.  Used for the timed CPU baseline
//                and for "is the GPU image statistically the reference's".
//   RngCounter — the GPU backend's counter-based generator keyed by
//                (seed, pixel, sample, bounce, stream).  With it a path
//                consumes exactly the numbers the GPU path consumes, so the
//                two can be compared ray by ray and pixel by pixel.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace orc {

static constexpr float INF = std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------------------
// src/rng.rs:11-60 — xoshiro256++
// ---------------------------------------------------------------------------
struct Seed {
    uint64_t s[4];
    static uint64_t rotl(uint64_t x, uint64_t k) { return (x << k) | (x >> (64 - k)); }

    // src/rng.rs:23-39
    uint64_t next() {
        uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    // src/rng.rs:43-53: top 53 bits -> f64 in [0,1) -> rounded to f32
    float next_f32() {
        const double scale = 1.0 / (double)(1ull << 53);
        uint64_t value = next() >> (64 - 53);
        return (float)(scale * (double)value);
    }
    // src/rng.rs:57-59
    float next_f32_range(float lo, float hi) { return lo + (hi - lo) * next_f32(); }
};

// ---------------------------------------------------------------------------
// Counter-based generator of the GPU backend (no counterpart in the reference:
// it takes the place of src/rng.rs's serial stream per image row, whose draws
// depend on the order in which rays are processed).  State is a pure function of
// the key; draws within one (bounce, stream) are sequential xoroshiro64**.
// ---------------------------------------------------------------------------
struct CounterStream {
    uint32_t s0, s1;
    static uint64_t mix64(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    static uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    static uint64_t path_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
        return mix64(seed + 0x9E3779B97F4A7C15ull * (((uint64_t)pixel << 32) | sample));
    }
    void init(uint64_t pkey, uint32_t bounce, uint32_t stream) {
        uint64_t b = mix64(pkey ^ (0xD1B54A32D192ED03ull * ((((uint64_t)bounce << 32) | stream) + 1)));
        s0 = (uint32_t)b;
        s1 = (uint32_t)(b >> 32);
        if ((s0 | s1) == 0) s0 = 0x9E3779B9u;
    }
    uint32_t next_u32() {
        uint32_t r = rotl32(s0 * 0x9E3779BBu, 5) * 5u;
        s1 ^= s0;
        s0 = rotl32(s0, 26) ^ s1 ^ (s1 << 9);
        s1 = rotl32(s1, 13);
        return r;
    }
    float next_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
};

enum : uint32_t { STREAM_CAMERA = 0, STREAM_SCATTER = 1, STREAM_MEDIUM0 = 2 };

// The reference threads one `&mut Seed` through get_ray, hit_anything (medium
// free-path draw) and scatter.  The two policies expose those three uses.
struct RngXoshiro {
    Seed seed;
    void begin_path(uint32_t, uint32_t) {}
    void begin_bounce(uint32_t) {}
    float camera() { return seed.next_f32(); }
    float scatter() { return seed.next_f32(); }
    float medium(uint32_t) { return seed.next_f32(); }
};

struct RngCounter {
    uint64_t seed = 0;
    uint64_t pkey = 0;
    uint32_t bounce = 0;
    CounterStream cam, sc;
    void begin_path(uint32_t pixel, uint32_t sample) {
        pkey = CounterStream::path_key(seed, pixel, sample);
        cam.init(pkey, 0, STREAM_CAMERA);
    }
    void begin_bounce(uint32_t b) {
        bounce = b;
        sc.init(pkey, b, STREAM_SCATTER);
    }
    float camera() { return cam.next_f32(); }
    float scatter() { return sc.next_f32(); }
    float medium(uint32_t medium_index) {
        CounterStream m;
        m.init(pkey, bounce, STREAM_MEDIUM0 + medium_index);
        return m.next_f32();
    }
};

// ---------------------------------------------------------------------------
// src/math/vec3.rs — f32x4-backed 3-vector, lane 3 == 0.  `reduce_sum` is
// LLVM's ordered reduction: ((x + y) + z) + w.
// ---------------------------------------------------------------------------
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
// src/math/vec3.rs:222-228: v / s == (1/s) * v
inline Vec3 operator/(Vec3 a, float s) { return (1.0f / s) * a; }
// src/math/vec3.rs:97-107
inline float dot(Vec3 a, Vec3 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + 0.0f; }
inline float length_squared(Vec3 a) { return dot(a, a); }
inline float length(Vec3 a) { return std::sqrt(length_squared(a)); }
// src/math/vec3.rs:110-119
inline Vec3 cross(Vec3 l, Vec3 r) {
    return {l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x};
}
// src/math/vec3.rs:123-125
inline Vec3 unit(Vec3 a) { return a / length(a); }
// src/math/vec3.rs:78-81
inline bool near_zero(Vec3 a) {
    const float T = 1e-8f;
    return std::fabs(a.x) < T && std::fabs(a.y) < T && std::fabs(a.z) < T;  // lane 3: |0| < T
}
// src/math/vec3.rs:84-86
inline Vec3 reflect(Vec3 v, Vec3 n) { return v - (2.0f * dot(v, n)) * n; }
// src/math/vec3.rs:89-94
inline Vec3 refract(Vec3 v, Vec3 n, float etai_over_etat) {
    float cos_theta = std::fmin(dot(-v, n), 1.0f);
    Vec3 r_out_perp = etai_over_etat * (v + cos_theta * n);
    Vec3 r_out_parallel = (-std::sqrt(std::fabs(1.0f - length_squared(r_out_perp)))) * n;
    return r_out_perp + r_out_parallel;
}
// src/math/vec3.rs:57-67 (three draws per trial, `< 1.0` accepts)
template <class F>
inline Vec3 random_in_unit_sphere(F&& next) {
    for (;;) {
        float a = -1.0f + (1.0f - -1.0f) * next();
        float b = -1.0f + (1.0f - -1.0f) * next();
        float c = -1.0f + (1.0f - -1.0f) * next();
        Vec3 p(a, b, c);
        if (length_squared(p) < 1.0f) return p;
    }
}
template <class F>
inline Vec3 random_unit(F&& next) { return unit(random_in_unit_sphere(next)); }
// src/math/vec3.rs:48-54
template <class F>
inline Vec3 random_in_unit_disk(F&& next) {
    for (;;) {
        float a = -1.0f + 2.0f * next();
        float b = -1.0f + 2.0f * next();
        Vec3 p(a, b, 0.0f);
        if (length_squared(p) < 1.0f) return p;
    }
}
// f32::to_radians: value * (PI / 180) with the constant rounded to f32
inline float to_radians(float deg) { return deg * 0.017453292519943295769236907684886f; }

// src/math/vec3.rs:252-259 — Rust `as u32` saturates and maps NaN to 0
inline uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
inline int32_t sat_i32(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
inline int64_t sat_i64(float f) {
    if (f != f) return 0;
    if (f >= 9223372036854775808.0f) return INT64_MAX;
    if (f <= -9223372036854775808.0f) return INT64_MIN;
    return (int64_t)f;
}
inline uint32_t to_rgba(Vec3 c) {
    float r = std::sqrt(c.x) * 255.999f, g = std::sqrt(c.y) * 255.999f, b = std::sqrt(c.z) * 255.999f;
    return (sat_u32(r) << 0) | (sat_u32(g) << 8) | (sat_u32(b) << 16);
}

// ---------------------------------------------------------------------------
// src/math/interval.rs
// ---------------------------------------------------------------------------
struct Interval {
    float min, max;
    bool contains(float x) const { return min <= x && x <= max; }   // :27-29 closed
    bool surrounds(float x) const { return min < x && x < max; }    // :33-35 open
    float size() const { return std::fabs(max - min); }             // :44-46
    float clamp(float x) const { return x < min ? min : (x > max ? max : x); }
};
static const Interval IV_EMPTY{INF, -INF};
static const Interval IV_UNIVERSE{-INF, INF};

// Rust f32::min/max and simd_min/simd_max return the non-NaN operand.
inline float fmin_(float a, float b) { return std::fmin(a, b); }
inline float fmax_(float a, float b) { return std::fmax(a, b); }

// ---------------------------------------------------------------------------
// src/math/aabb.rs — lane 3 of mins/maxs is 1.0 (the ray_t trick)
// ---------------------------------------------------------------------------
struct AABB {
    float mins[4], maxs[4];
    Interval axis_interval(int a) const { return {mins[a], maxs[a]}; }
    // :79-98
    void pad_to_minimums() {
        const float delta = 0.0001f;
        const float delta_half = delta * 0.5f;
        for (int a = 0; a < 3; ++a) {
            if (axis_interval(a).size() < delta) {
                mins[a] = mins[a] - delta_half;
                maxs[a] = maxs[a] + delta_half;
            }
        }
    }
    // :22-30
    static AABB make(Interval x, Interval y, Interval z) {
        AABB b{{x.min, y.min, z.min, 1.0f}, {x.max, y.max, z.max, 1.0f}};
        b.pad_to_minimums();
        return b;
    }
    static AABB empty() { return make(IV_EMPTY, IV_EMPTY, IV_EMPTY); }  // :20
    bool operator==(const AABB& o) const {
        for (int i = 0; i < 4; ++i)
            if (!(mins[i] == o.mins[i]) || !(maxs[i] == o.maxs[i])) return false;
        return true;
    }
    // :52-58
    static AABB from_points(Vec3 a, Vec3 b) {
        Interval iv[3];
        for (int i = 0; i < 3; ++i) iv[i] = a[i] <= b[i] ? Interval{a[i], b[i]} : Interval{b[i], a[i]};
        return make(iv[0], iv[1], iv[2]);
    }
    // :61-69 (+ src/math/interval.rs:18-23)
    static AABB from_aabbs(const AABB& b1, const AABB& b2) {
        if (b2 == empty()) return b1;
        Interval iv[3];
        for (int i = 0; i < 3; ++i)
            iv[i] = Interval{fmin_(b1.mins[i], b2.mins[i]), fmax_(b1.maxs[i], b2.maxs[i])};
        return make(iv[0], iv[1], iv[2]);
    }
    // :73-76
    int longest_axis() const {
        float sx = axis_interval(0).size(), sy = axis_interval(1).size(), sz = axis_interval(2).size();
        if (sx > sy) return sx > sz ? 0 : 2;
        return sy > sz ? 1 : 2;
    }
    // :109-111 (lane 3 of a Vec3 is 0, so lane 3 stays 1)
    AABB offset(Vec3 o) const {
        AABB r = *this;
        for (int i = 0; i < 3; ++i) { r.mins[i] = mins[i] + o[i]; r.maxs[i] = maxs[i] + o[i]; }
        return r;
    }
};

struct Ray {
    Vec3 origin, direction;
    float time = 0;
    Vec3 at(float t) const { return origin + t * direction; }  // src/math/ray.rs:55
};

// src/math/aabb.rs:115-186
struct AABBx2 {
    float mins[8], maxs[8];
    static AABBx2 make(const AABB& a, const AABB& b) {
        AABBx2 r;
        for (int i = 0; i < 4; ++i) { r.mins[i] = a.mins[i]; r.mins[4 + i] = b.mins[i]; r.maxs[i] = a.maxs[i]; r.maxs[4 + i] = b.maxs[i]; }
        return r;
    }
    AABB aabb1() const { return AABB::make({mins[0], maxs[0]}, {mins[1], maxs[1]}, {mins[2], maxs[2]}); }
    AABB aabb2() const { return AABB::make({mins[4], maxs[4]}, {mins[5], maxs[5]}, {mins[6], maxs[6]}); }
    // :127-167 — both child boxes; lane 3 carries ray_t through the reductions
    void hit(const Ray& ray, Interval ray_t, Interval out_t[2], bool out_hit[2]) const {
        float inv[3] = {1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z};
        float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z};
        for (int c = 0; c < 2; ++c) {
            const float* mn = mins + 4 * c;
            const float* mx = maxs + 4 * c;
            float t1[4], t2[4];
            for (int i = 0; i < 3; ++i) { t1[i] = (mn[i] - o[i]) * inv[i]; t2[i] = (mx[i] - o[i]) * inv[i]; }
            t1[3] = (mn[3] - 0.0f) * ray_t.min;
            t2[3] = (mx[3] - 0.0f) * ray_t.max;
            float lo = fmin_(t1[0], t2[0]), hi = fmax_(t1[0], t2[0]);
            for (int i = 1; i < 4; ++i) { lo = fmax_(lo, fmin_(t1[i], t2[i])); hi = fmin_(hi, fmax_(t1[i], t2[i])); }
            out_t[c] = Interval{lo, hi};
            out_hit[c] = lo <= hi;
        }
    }
};

// ---------------------------------------------------------------------------
// src/perlin_noise.rs
// ---------------------------------------------------------------------------
struct PerlinNoise {
    std::vector<Vec3> rand_floats;
    std::vector<uint64_t> perm_x, perm_y, perm_z;

    // :117-122
    static void permute(Seed& seed, std::vector<uint64_t>& p, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            size_t target = (size_t)(seed.next() % (uint64_t)(i + 1));
            std::swap(p[i], p[target]);
        }
    }
    // :106-114
    static void generate_perm(Seed& seed, std::vector<uint64_t>& p, size_t n) {
        p.clear();
        for (size_t i = 0; i < n; ++i) p.push_back(i);
        permute(seed, p, n);
    }
    // :17-37
    static PerlinNoise make(Seed& seed, size_t point_count) {
        PerlinNoise pn;
        auto next = [&]() { return seed.next_f32(); };
        for (size_t i = 0; i < point_count; ++i) pn.rand_floats.push_back(random_unit(next));
        generate_perm(seed, pn.perm_x, point_count);
        generate_perm(seed, pn.perm_y, point_count);
        generate_perm(seed, pn.perm_z, point_count);
        return pn;
    }
    // :39-87
    float noise(Vec3 p) const {
        float fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
        float u = p.x - fx, v = p.y - fy, w = p.z - fz;
        int64_t i = sat_i64(fx), j = sat_i64(fy), k = sat_i64(fz);
        size_t n = rand_floats.size();
        Vec3 c[2][2][2];
        for (int64_t di = 0; di < 2; ++di)
            for (int64_t dj = 0; dj < 2; ++dj)
                for (int64_t dk = 0; dk < 2; ++dk)
                    // (i + di) wraps in the reference's release build; unsigned arithmetic says so without overflow
                    c[di][dj][dk] = rand_floats[(perm_x[(size_t)(((uint64_t)i + (uint64_t)di) & 255) % n] ^
                                                 perm_y[(size_t)(((uint64_t)j + (uint64_t)dj) & 255) % n] ^
                                                 perm_z[(size_t)(((uint64_t)k + (uint64_t)dk) & 255) % n]) % n];
        float uu = u * u * (3.0f - 2.0f * u);
        float vv = v * v * (3.0f - 2.0f * v);
        float ww = w * w * (3.0f - 2.0f * w);
        float accum = 0.0f;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int d = 0; d < 2; ++d) {
                    Vec3 weight_v(u - (float)a, v - (float)b, w - (float)d);
                    accum += ((float)a * uu + (1.0f - (float)a) * (1.0f - uu)) *
                             ((float)b * vv + (1.0f - (float)b) * (1.0f - vv)) *
                             ((float)d * ww + (1.0f - (float)d) * (1.0f - ww)) *
                             dot(c[a][b][d], weight_v);
                }
        return accum;
    }
    // :90-102
    float turbulance(Vec3 p, size_t depth) const {
        float accum = 0.0f;
        Vec3 temp_p = p;
        float weight = 1.0f;
        for (size_t i = 0; i < depth; ++i) {
            accum += weight * noise(temp_p);
            weight *= 0.5f;
            temp_p = 2.0f * temp_p;
        }
        return std::fabs(accum);
    }
};

// ---------------------------------------------------------------------------
// src/texture.rs
// ---------------------------------------------------------------------------
struct Image {  // image::Rgb32FImage: f32 RGB, row-major, top row first
    uint32_t width = 0, height = 0;
    std::vector<float> rgb;
};

struct Texture {
    enum Kind { SOLID, CHECKER, IMAGE, NOISE, NOT_FOUND } kind = NOT_FOUND;
    Vec3 colour;
    float inv_scale = 0;          // checker
    const Texture* even = nullptr;
    const Texture* odd = nullptr;
    const Image* image = nullptr;
    const PerlinNoise* noise = nullptr;
    float scale = 0;              // noise
    uint32_t id = 0;              // oracle bookkeeping (creation index)

    // :40-73
    Vec3 value(float u, float v, Vec3 p) const {
        switch (kind) {
            case SOLID: return colour;
            case CHECKER: {
                float fx = std::floor(inv_scale * p.x), fy = std::floor(inv_scale * p.y), fz = std::floor(inv_scale * p.z);
                float f3 = std::floor(inv_scale * 0.0f);
                float sum = ((fx + fy) + fz) + f3;
                bool is_even = sat_i32(sum) % 2 == 0;
                return (is_even ? even : odd)->value(u, v, p);
            }
            case IMAGE: {
                Interval unit_iv{0.0f, 1.0f};
                u = unit_iv.clamp(u);
                v = 1.0f - unit_iv.clamp(v);
                uint32_t i = sat_u32(u * (float)(image->width - 1));
                uint32_t j = sat_u32(v * (float)(image->height - 1));
                const float* px = &image->rgb[3 * ((size_t)j * image->width + i)];
                return Vec3(px[0] * px[0], px[1] * px[1], px[2] * px[2]);
            }
            case NOISE:
                return (1.0f + std::sin(scale * p.z + 10.0f * noise->turbulance(p, 7))) * Vec3(0.5f, 0.5f, 0.5f);
            default: return Vec3(0, 0, 0);
        }
    }
};

// ---------------------------------------------------------------------------
// src/hittable.rs:7-26
// ---------------------------------------------------------------------------
struct HitRecord {
    Vec3 point, normal;
    float t = 0;
    bool front_face = false;
    uint32_t material = 0;
    float u = 0, v = 0;
    uint32_t prim_id = 0xFFFFFFFFu;  // oracle bookkeeping, not in the reference
    void set_face_normal(const Ray& ray, Vec3 outward_normal) {
        front_face = dot(ray.direction, outward_normal) < 0.0f;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

// ---------------------------------------------------------------------------
// src/material.rs
// ---------------------------------------------------------------------------
struct Material {
    enum Kind { LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT, ISOTROPIC, NOT_FOUND } kind = NOT_FOUND;
    float param = 0;  // fuzz_radius | refraction_index
    const Texture* texture = nullptr;

    // :155-160
    static float reflectance(float cos, float rr) {
        float r0 = (1.0f - rr) / (1.0f + rr);
        r0 = r0 * r0;
        float x = 1.0f - cos;
        float x2 = x * x;
        return r0 + (1.0f - r0) * (x * (x2 * x2));  // powi(5)
    }
    // :117-122
    Vec3 emitted(float u, float v, Vec3 p) const {
        return kind == DIFFUSE_LIGHT ? texture->value(u, v, p) : Vec3(0, 0, 0);
    }
    // :65-114 — returns false for None
    template <class R>
    bool scatter(R& rng, const Ray& ray_in, const HitRecord& rec, Ray& scattered, Vec3& attenuation) const {
        auto next = [&]() { return rng.scatter(); };
        switch (kind) {
            case LAMBERTIAN: {
                Vec3 scatter_dir = rec.normal + random_unit(next);
                if (near_zero(scatter_dir)) scatter_dir = rec.normal;
                scattered = Ray{rec.point, scatter_dir, ray_in.time};
                attenuation = texture->value(rec.u, rec.v, rec.point);
                return true;
            }
            case METAL: {
                float fuzz_radius = fmin_(param, 1.0f);
                Vec3 reflected = reflect(unit(ray_in.direction), rec.normal);
                scattered = Ray{rec.point, reflected + fuzz_radius * random_unit(next), ray_in.time};
                if (dot(scattered.direction, rec.normal) > 0.0f) {
                    attenuation = texture->value(rec.u, rec.v, rec.point);
                    return true;
                }
                return false;
            }
            case DIELECTRIC: {
                attenuation = texture->value(rec.u, rec.v, rec.point);
                float refraction_ratio = rec.front_face ? 1.0f / param : param;
                Vec3 unit_dir = unit(ray_in.direction);
                float cos_theta = fmin_(dot(-unit_dir, rec.normal), 1.0f);
                float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
                bool cannot_refract = refraction_ratio * sin_theta > 1.0f;
                Vec3 direction;
                if (cannot_refract || reflectance(cos_theta, refraction_ratio) > next())
                    direction = reflect(unit_dir, rec.normal);
                else
                    direction = refract(unit_dir, rec.normal, refraction_ratio);
                scattered = Ray{rec.point, direction, ray_in.time};
                return true;
            }
            case ISOTROPIC: {
                scattered = Ray{rec.point, random_unit(next), ray_in.time};
                attenuation = texture->value(rec.u, rec.v, rec.point);
                return true;
            }
            case DIFFUSE_LIGHT: return false;
            default: return false;  // NotFound: unimplemented!() in the reference
        }
    }
};

// src/material.rs:34-49 — id 0 is a black lambertian
struct MaterialMap {
    std::vector<Material> map;
};

// ---------------------------------------------------------------------------
// src/hittable.rs — leaves
// ---------------------------------------------------------------------------
static constexpr float PI_F = 3.14159265358979323846264338327950288f;

// :341-352
inline void get_sphere_uv(Vec3 p, float& u, float& v) {
    float theta = std::acos(-p.y);
    float phi = std::atan2(-p.z, p.x) + PI_F;
    u = phi / (2.0f * PI_F);
    v = theta / PI_F;
}

struct Counters {
    uint64_t samples = 0, segments = 0, node_visits = 0, sphere_tests = 0, quad_tests = 0, medium_tests = 0, hits = 0;
    void add(const Counters& o) {
        samples += o.samples; segments += o.segments; node_visits += o.node_visits; sphere_tests += o.sphere_tests;
        quad_tests += o.quad_tests; medium_tests += o.medium_tests; hits += o.hits;
    }
};

struct Sphere {
    Vec3 centre; float radius; uint32_t material; uint32_t prim_id;
    // :262-288
    bool hit(const Ray& ray, Interval t, HitRecord& rec) const {
        Vec3 oc = ray.origin - centre;
        float a = length_squared(ray.direction);
        float half_b = dot(oc, ray.direction);
        float c = length_squared(oc) - radius * radius;
        float discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0f) return false;
        float ds = std::sqrt(discriminant);
        float root = (-half_b - ds) / a;
        if (!t.surrounds(root)) {
            root = (-half_b + ds) / a;
            if (!t.surrounds(root)) return false;
        }
        rec.t = root;
        rec.point = ray.at(rec.t);
        Vec3 outward_normal = (rec.point - centre) / radius;
        rec.set_face_normal(ray, outward_normal);
        get_sphere_uv(outward_normal, rec.u, rec.v);
        rec.material = material;
        rec.prim_id = prim_id;
        return true;
    }
};

struct MovingSphere {
    Ray centre; float radius; uint32_t material; uint32_t prim_id;
    // :309-336
    bool hit(const Ray& ray, Interval t, HitRecord& rec) const {
        Vec3 current_centre = centre.at(ray.time);
        Vec3 oc = ray.origin - current_centre;
        float a = length_squared(ray.direction);
        float half_b = dot(oc, ray.direction);
        float c = length_squared(oc) - radius * radius;
        float discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0f) return false;
        float ds = std::sqrt(discriminant);
        float root = (-half_b - ds) / a;
        if (!t.surrounds(root)) {
            root = (-half_b + ds) / a;
            if (!t.surrounds(root)) return false;
        }
        rec.t = root;
        rec.point = ray.at(rec.t);
        Vec3 outward_normal = (rec.point - current_centre) / radius;
        rec.set_face_normal(ray, outward_normal);
        get_sphere_uv(outward_normal, rec.u, rec.v);
        rec.material = material;
        rec.prim_id = prim_id;
        return true;
    }
};

struct Quad {
    Vec3 q, u, v, n; uint32_t material; uint32_t prim_id;
    // :365-368
    static Quad make(Vec3 q, Vec3 u, Vec3 v, uint32_t material, uint32_t prim_id) {
        return Quad{q, u, v, cross(u, v), material, prim_id};
    }
    // :372-408
    bool hit(const Ray& ray, Interval ray_t, HitRecord& rec) const {
        Vec3 normal = unit(n);
        float d = dot(normal, q);
        Vec3 w = n / dot(n, n);
        float denom = dot(normal, ray.direction);
        if (std::fabs(denom) < 1e-8f) return false;
        float t = (d - dot(normal, ray.origin)) / denom;
        if (!ray_t.contains(t)) return false;
        Vec3 intersection = ray.at(t);
        Vec3 planar_hitpt_vec = intersection - q;
        float alpha = dot(w, cross(planar_hitpt_vec, v));
        float beta = dot(w, cross(u, planar_hitpt_vec));
        Interval unit_iv{0.0f, 1.0f};
        if (!unit_iv.contains(alpha) || !unit_iv.contains(beta)) return false;
        rec.t = t;
        rec.u = alpha;
        rec.v = beta;
        rec.point = intersection;
        rec.material = material;
        rec.prim_id = prim_id;
        rec.set_face_normal(ray, normal);
        return true;
    }
};

struct Hittable;

struct ConstantMedium {
    uint32_t phase_function; const Hittable* boundary; float neg_inv_density;
    uint32_t medium_index;  // creation order among media: key of the counter-RNG medium stream
    uint32_t prim_id;
};

// src/hittable.rs:29-57
struct Hittable {
    enum Kind : uint32_t { SPHERE, MOVING_SPHERE, QUAD, CONSTANT_MEDIUM, BVH, MOVE, ROTATE_Y } kind;
    union {
        Sphere sphere;
        MovingSphere moving;
        Quad quad;
        ConstantMedium medium;
        struct { const AABBx2* aabbs; const Hittable* left; const Hittable* right; } bvh;
        struct { const Hittable* obj; Vec3 offset; } move;
        struct { const Hittable* obj; float sin, cos; } rot;
    };
    Hittable() : kind(SPHERE), sphere{} {}
    Hittable(const Hittable& o) { std::memcpy((void*)this, (const void*)&o, sizeof(Hittable)); }
    Hittable& operator=(const Hittable& o) { std::memcpy((void*)this, (const void*)&o, sizeof(Hittable)); return *this; }

    // :158-224
    AABB calc_aabb() const {
        switch (kind) {
            case SPHERE: {
                Vec3 rvec(sphere.radius, sphere.radius, sphere.radius);
                return AABB::from_points(sphere.centre - rvec, sphere.centre + rvec);
            }
            case MOVING_SPHERE: {
                Vec3 rvec(moving.radius, moving.radius, moving.radius);
                AABB b1 = AABB::from_points(moving.centre.at(0.0f) - rvec, moving.centre.at(0.0f) + rvec);
                AABB b2 = AABB::from_points(moving.centre.at(1.0f) - rvec, moving.centre.at(1.0f) + rvec);
                return AABB::from_aabbs(b1, b2);
            }
            case QUAD: {
                AABB d1 = AABB::from_points(quad.q, quad.q + quad.u + quad.v);
                AABB d2 = AABB::from_points(quad.q + quad.u, quad.q + quad.v);
                return AABB::from_aabbs(d1, d2);
            }
            case BVH: return AABB::from_aabbs(bvh.aabbs->aabb1(), bvh.aabbs->aabb2());
            case CONSTANT_MEDIUM: return medium.boundary->calc_aabb();
            case MOVE: return move.obj->calc_aabb().offset(move.offset);
            case ROTATE_Y: {
                Vec3 mn(INF, INF, INF), mx(-INF, -INF, -INF);
                AABB bbox = rot.obj->calc_aabb();
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        for (int k = 0; k < 2; ++k) {
                            float fi = (float)i, fj = (float)j, fk = (float)k;
                            float x = fi * bbox.maxs[0] + (1.0f - fi) * bbox.mins[0];
                            float y = fj * bbox.maxs[1] + (1.0f - fj) * bbox.mins[1];
                            float z = fk * bbox.maxs[2] + (1.0f - fk) * bbox.mins[2];
                            float newx = rot.cos * x + rot.sin * z;
                            float newz = -rot.sin * x + rot.cos * z;
                            Vec3 tester(newx, y, newz);
                            for (int c = 0; c < 3; ++c) { mn[c] = fmin_(mn[c], tester[c]); mx[c] = fmax_(mx[c], tester[c]); }
                        }
                return AABB::from_points(mn, mx);
            }
        }
        return AABB::empty();
    }
};

// ---------------------------------------------------------------------------
// Scene: owner of everything the reference puts in its Arena, plus the camera.
// ---------------------------------------------------------------------------
struct RaytracingCamera {  // src/camera.rs:206-218
    size_t width = 0, height = 0;
    Vec3 centre, pixel00_loc, pixel_delta_u, pixel_delta_v;
    size_t max_depth = 0;
    float defocus_angle = 0;
    Vec3 defocus_disk_u, defocus_disk_v;
    Vec3 background_colour;

    // src/camera.rs:222-266
    static RaytracingCamera make(size_t width, size_t height, size_t max_depth, float vfov, Vec3 look_from, Vec3 look_at,
                                 Vec3 vup, float defocus_angle, float focus_dist, Vec3 background_colour) {
        RaytracingCamera c;
        Vec3 centre = look_from;
        float theta = to_radians(vfov);
        float h = std::tan(theta / 2.0f);
        float viewport_height = 2.0f * h * focus_dist;
        float viewport_width = viewport_height * ((float)width / (float)height);
        Vec3 w = unit(look_from - look_at);
        Vec3 u = unit(cross(vup, w));
        Vec3 v = cross(w, u);
        Vec3 viewport_u = viewport_width * u;
        Vec3 viewport_v = viewport_height * -v;
        Vec3 pixel_delta_u = viewport_u / (float)width;
        Vec3 pixel_delta_v = viewport_v / (float)height;
        Vec3 viewport_upper_left = centre - (focus_dist * w) - viewport_u / 2.0f - viewport_v / 2.0f;
        Vec3 pixel00_loc = viewport_upper_left + 0.5f * (pixel_delta_u + pixel_delta_v);
        float defocus_radius = focus_dist * std::tan(to_radians(defocus_angle / 2.0f));
        c.width = width; c.height = height; c.centre = centre; c.pixel00_loc = pixel00_loc;
        c.pixel_delta_u = pixel_delta_u; c.pixel_delta_v = pixel_delta_v; c.max_depth = max_depth;
        c.defocus_angle = defocus_angle; c.defocus_disk_u = defocus_radius * u; c.defocus_disk_v = defocus_radius * v;
        c.background_colour = background_colour;
        return c;
    }
};

struct Scene {
    std::deque<Hittable> hittables;      // Arena-allocated Hittable nodes
    std::deque<AABBx2> aabbs;
    std::deque<Texture> textures;
    std::deque<Image> images;
    std::deque<PerlinNoise> perlins;
    MaterialMap mmap;
    const Hittable* entry = nullptr;     // World::entry, src/lib.rs:17-26
    uint32_t next_prim_id = 0;
    uint32_t next_medium = 0;
    uint32_t bvh_assert_failures = 0;

    // Camera state (src/camera.rs:10-32)
    Vec3 position, vup, background;
    float pitch = 0, yaw = 0, vfov = 0, focus_dist = 0, defocus_angle = 0;
    size_t width = 0, height = 0, max_depth = 0;
    RaytracingCamera rt_cam;

    Scene() {
        // MaterialMap::new, src/material.rs:40-44
        Texture t; t.kind = Texture::SOLID; t.colour = Vec3(0, 0, 0); t.id = 0;
        textures.push_back(t);
        mmap.map.push_back(Material{Material::LAMBERTIAN, 0.0f, &textures.back()});
    }

    const Hittable* alloc(const Hittable& h) { hittables.push_back(h); return &hittables.back(); }

    const Texture* tex_colour(Vec3 c) { Texture t; t.kind = Texture::SOLID; t.colour = c; return push_tex(t); }
    // src/texture.rs:85-87
    const Texture* tex_checker(float scale, const Texture* even, const Texture* odd) {
        Texture t; t.kind = Texture::CHECKER; t.inv_scale = 1.0f / scale; t.even = even; t.odd = odd; return push_tex(t);
    }
    const Texture* tex_image(const Image* im) { Texture t; t.kind = Texture::IMAGE; t.image = im; return push_tex(t); }
    const Texture* tex_noise(const PerlinNoise* pn, float scale) { Texture t; t.kind = Texture::NOISE; t.noise = pn; t.scale = scale; return push_tex(t); }
    const Texture* push_tex(Texture t) { t.id = (uint32_t)textures.size(); textures.push_back(t); return &textures.back(); }

    uint32_t push_material(Material::Kind k, const Texture* t, float param = 0.0f) {
        mmap.map.push_back(Material{k, param, t});
        return (uint32_t)mmap.map.size() - 1;
    }

    Hittable sphere(Vec3 c, float r, uint32_t mat) {
        Hittable h; h.kind = Hittable::SPHERE; h.sphere = Sphere{c, r, mat, next_prim_id++}; return h;
    }
    // src/hittable.rs:302-306
    Hittable moving_sphere(Vec3 c1, Vec3 c2, float r, uint32_t mat) {
        Hittable h; h.kind = Hittable::MOVING_SPHERE;
        h.moving = MovingSphere{Ray{c1, c2 - c1, 0.0f}, r, mat, next_prim_id++}; return h;
    }
    Hittable quad(Vec3 q, Vec3 u, Vec3 v, uint32_t mat) {
        Hittable h; h.kind = Hittable::QUAD; h.quad = Quad::make(q, u, v, mat, next_prim_id++); return h;
    }
    // src/hittable.rs:419-425
    Hittable constant_medium(const Hittable* boundary, float density, const Texture* tex) {
        uint32_t phase = push_material(Material::ISOTROPIC, tex);
        Hittable h; h.kind = Hittable::CONSTANT_MEDIUM;
        h.medium = ConstantMedium{phase, boundary, -1.0f / density, next_medium++, next_prim_id++};
        return h;
    }
    // src/hittable.rs:227-231
    Hittable move_by(const Hittable& h, Vec3 offset) {
        Hittable m; m.kind = Hittable::MOVE; m.move.obj = alloc(h); m.move.offset = offset; return m;
    }
    // src/hittable.rs:234-243
    Hittable rotate_y_by(const Hittable& h, float degrees) {
        float rads = to_radians(degrees);
        Hittable m; m.kind = Hittable::ROTATE_Y; m.rot.obj = alloc(h); m.rot.sin = std::sin(rads); m.rot.cos = std::cos(rads); return m;
    }
    // src/hittable.rs:81-100
    Hittable box_of_quads(Vec3 a, Vec3 b, uint32_t mat) {
        Vec3 mn(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z));
        Vec3 mx(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z));
        Vec3 dx(mx.x - mn.x, 0, 0), dy(0, mx.y - mn.y, 0), dz(0, 0, mx.z - mn.z);
        std::vector<Hittable> sides = {
            quad(Vec3(mn.x, mn.y, mx.z), dx, dy, mat),    // front
            quad(Vec3(mx.x, mn.y, mx.z), -dz, dy, mat),   // right
            quad(Vec3(mx.x, mn.y, mn.z), -dx, dy, mat),   // back
            quad(Vec3(mn.x, mn.y, mn.z), dz, dy, mat),    // left
            quad(Vec3(mn.x, mx.y, mx.z), dx, -dz, mat),   // top
            quad(Vec3(mn.x, mn.y, mn.z), dx, dz, mat),    // bottom
        };
        return bvh(sides);
    }
    // src/hittable.rs:103-155.  slice::sort_by only asks "is a < b"; with the
    // reference's comparator (Less iff a.min < b.min) that is a stable sort on
    // the AABB minimum along the longest axis.
    Hittable bvh(const std::vector<Hittable>& hittables_in);

    // src/camera.rs:36-63,133-145,149-160
    void camera_new(Vec3 pos, size_t w, size_t h, size_t depth, float vfov_, Vec3 vup_, float defocus, float focus, Vec3 bg) {
        position = pos; width = w; height = h; max_depth = depth; vfov = vfov_; vup = vup_;
        defocus_angle = defocus; focus_dist = focus; background = bg; pitch = 0; yaw = 0;
        update_raytracing_camera();
    }
    void change_pitch_yaw_by(float dp, float dy) { pitch += dp; yaw += dy; update_raytracing_camera(); }
    void update_raytracing_camera() {
        Vec3 direction(std::cos(to_radians(yaw)) * std::cos(to_radians(pitch)), std::sin(to_radians(pitch)),
                       std::sin(to_radians(yaw)) * std::cos(to_radians(pitch)));
        rt_cam = RaytracingCamera::make(width, height, max_depth, vfov, position, position + direction, vup,
                                        defocus_angle, focus_dist, background);
    }
};

inline Hittable Scene::bvh(const std::vector<Hittable>& in) {
    AABB aabb = AABB::make(IV_EMPTY, IV_EMPTY, IV_EMPTY);
    for (const Hittable& l : in) aabb = AABB::from_aabbs(aabb, l.calc_aabb());
    int axis = aabb.longest_axis();
    const Hittable* r1;
    const Hittable* r2 = nullptr;
    if (in.size() == 1) {
        r1 = alloc(in[0]);
    } else if (in.size() == 2) {
        r1 = alloc(in[0]);
        r2 = alloc(in[1]);
    } else {
        // calc_aabb is recomputed inside the reference's comparator; cache the key (same value)
        std::vector<std::pair<float, size_t>> keys(in.size());
        for (size_t i = 0; i < in.size(); ++i) keys[i] = {in[i].calc_aabb().axis_interval(axis).min, i};
        std::stable_sort(keys.begin(), keys.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        size_t middle = in.size() / 2;
        std::vector<Hittable> lo, hi;
        for (size_t i = 0; i < in.size(); ++i) (i < middle ? lo : hi).push_back(in[keys[i].second]);
        r1 = alloc(bvh(lo));
        r2 = alloc(bvh(hi));
    }
    aabbs.push_back(AABBx2::make(r1->calc_aabb(), r2 ? r2->calc_aabb() : AABB::empty()));
    Hittable h;
    h.kind = Hittable::BVH;
    h.bvh.aabbs = &aabbs.back();
    h.bvh.left = r1;
    h.bvh.right = r2 ? r2 : r1;
    if (!(h.calc_aabb() == aabb)) ++bvh_assert_failures;  // assert_eq! at :153
    return h;
}

// ---------------------------------------------------------------------------
// src/math/ray.rs:16-44,59-275 — closest hit with explicit stacks
// ---------------------------------------------------------------------------
struct Switch { const Hittable* hittable; };  // nullptr == Switch::Modification

struct Modification {
    enum Kind { RAY_MOVE, RAY_ROTATE_Y, MEDIUM_PHASE1, MEDIUM_PHASE2 } kind;
    Vec3 vec; bool hit;                                   // RayMove
    Ray original_ray; bool hit_anything_prev; float sin, cos;  // RayRotateY (+ media)
    HitRecord original_rec; Interval original_t; HitRecord rec_1; const ConstantMedium* medium;
};

struct Stacks {
    std::vector<Switch> hittable_stack;
    std::vector<Modification> modification_stack;
    Counters counters;
};

template <class R>
inline bool hit_anything(const Ray& self, R& rng, HitRecord& rec, const Hittable* entry, Stacks& st,
                         float tmin0 = 0.001f, float tmax0 = INF) {
    auto& hs = st.hittable_stack;
    auto& ms = st.modification_stack;
    bool hit_any = false;
    float tmin = tmin0, tmax = tmax0;
    Ray ray = self;
    st.counters.segments++;
    hs.push_back(Switch{entry});

    while (!hs.empty()) {
        Switch sw = hs.back();
        hs.pop_back();
        if (sw.hittable == nullptr) {
            Modification m = ms.back();
            ms.pop_back();
            switch (m.kind) {
                case Modification::RAY_MOVE:  // :75-84
                    ray.origin = ray.origin + m.vec;
                    if (hit_any) rec.point = rec.point + m.vec;
                    hit_any = hit_any || m.hit;
                    continue;
                case Modification::RAY_ROTATE_Y:  // :86-105
                    ray = m.original_ray;
                    if (hit_any) {
                        rec.point = Vec3((m.cos * rec.point.x) + (m.sin * rec.point.z), rec.point.y,
                                         (-m.sin * rec.point.x) + (m.cos * rec.point.z));
                        rec.normal = Vec3((m.cos * rec.normal.x) + (m.sin * rec.normal.z), rec.normal.y,
                                          (-m.sin * rec.normal.x) + (m.cos * rec.normal.z));
                    }
                    hit_any = hit_any || m.hit_anything_prev;
                    continue;
                case Modification::MEDIUM_PHASE1: {  // :108-128
                    if (!hit_any) {
                        hit_any = m.hit_anything_prev;
                        rec = m.original_rec;
                        tmin = m.original_t.min;
                        tmax = m.original_t.max;
                        continue;
                    }
                    Modification p2 = m;
                    p2.kind = Modification::MEDIUM_PHASE2;
                    p2.rec_1 = rec;
                    ms.push_back(p2);
                    hs.push_back(Switch{nullptr});
                    hs.push_back(Switch{m.medium->boundary});
                    tmin = 0.0001f + rec.t;
                    tmax = IV_UNIVERSE.max;
                    rec = HitRecord();
                    hit_any = false;
                    continue;
                }
                case Modification::MEDIUM_PHASE2: {  // :131-163
                    HitRecord rec_2 = rec;
                    rec = m.original_rec;
                    HitRecord rec_1 = m.rec_1;
                    tmin = m.original_t.min;
                    tmax = m.original_t.max;
                    if (!hit_any) { hit_any = m.hit_anything_prev; continue; }
                    hit_any = m.hit_anything_prev;
                    if (rec_1.t < tmin) rec_1.t = tmin;
                    if (rec_2.t > tmax) rec_2.t = tmax;
                    if (rec_1.t >= rec_2.t) continue;
                    rec_1.t = fmax_(rec_1.t, 0.0f);
                    float ray_len = length(ray.direction);
                    float distance_inside_bounds = (rec_2.t - rec_1.t) * ray_len;
                    // f32::log(self, base) = ln(self) / ln(base), base = E as f32
                    float hit_distance = m.medium->neg_inv_density *
                                         (std::log(rng.medium(m.medium->medium_index)) / std::log((float)2.718281828459045));
                    if (hit_distance > distance_inside_bounds) continue;
                    rec.t = rec_1.t + hit_distance / ray_len;
                    tmax = rec.t;
                    rec.point = ray.at(rec.t);
                    rec.normal = Vec3(1.0f, 0.0f, 0.0f);
                    rec.front_face = true;
                    rec.material = m.medium->phase_function;
                    rec.prim_id = m.medium->prim_id;
                    hit_any = true;
                    continue;
                }
            }
            continue;
        }

        const Hittable* h = sw.hittable;
        Interval t{tmin, tmax};
        bool hit = false;
        switch (h->kind) {
            case Hittable::SPHERE: st.counters.sphere_tests++; hit = h->sphere.hit(ray, t, rec); break;   // :176
            case Hittable::QUAD: st.counters.quad_tests++; hit = h->quad.hit(ray, t, rec); break;         // :177
            case Hittable::MOVING_SPHERE: st.counters.sphere_tests++; hit = h->moving.hit(ray, t, rec); break;  // :178
            case Hittable::BVH: {  // :179-204
                st.counters.node_visits++;
                Interval bt[2]; bool bh[2];
                h->bvh.aabbs->hit(ray, t, bt, bh);
                if (bh[0] && bh[1]) {
                    if (bt[0].max <= bt[1].max) { hs.push_back(Switch{h->bvh.right}); hs.push_back(Switch{h->bvh.left}); }
                    else { hs.push_back(Switch{h->bvh.left}); hs.push_back(Switch{h->bvh.right}); }
                } else if (bh[0]) hs.push_back(Switch{h->bvh.left});
                else if (bh[1]) hs.push_back(Switch{h->bvh.right});
                continue;
            }
            case Hittable::MOVE: {  // :207-215
                ray.origin = ray.origin - h->move.offset;
                Modification m{}; m.kind = Modification::RAY_MOVE; m.vec = h->move.offset; m.hit = hit_any;
                ms.push_back(m);
                hs.push_back(Switch{nullptr});
                hs.push_back(Switch{h->move.obj});
                hit_any = false;
                continue;
            }
            case Hittable::ROTATE_Y: {  // :217-247 (hit_anything is NOT reset here, as in the reference)
                float s = h->rot.sin, c = h->rot.cos;
                Vec3 origin((c * ray.origin.x) - (s * ray.origin.z), ray.origin.y, (s * ray.origin.x) + (c * ray.origin.z));
                Vec3 direction((c * ray.direction.x) - (s * ray.direction.z), ray.direction.y,
                               (s * ray.direction.x) + (c * ray.direction.z));
                Modification m{}; m.kind = Modification::RAY_ROTATE_Y; m.original_ray = ray; m.hit_anything_prev = hit_any;
                m.sin = s; m.cos = c;
                ray.origin = origin;
                ray.direction = direction;
                ms.push_back(m);
                hs.push_back(Switch{nullptr});
                hs.push_back(Switch{h->rot.obj});
                continue;
            }
            case Hittable::CONSTANT_MEDIUM: {  // :250-265
                st.counters.medium_tests++;
                Modification m{}; m.kind = Modification::MEDIUM_PHASE1; m.original_rec = rec; m.original_t = t;
                m.medium = &h->medium; m.hit_anything_prev = hit_any;
                ms.push_back(m);
                hs.push_back(Switch{nullptr});
                hs.push_back(Switch{h->medium.boundary});
                rec = HitRecord();
                hit_any = false;
                tmin = IV_UNIVERSE.min;
                tmax = IV_UNIVERSE.max;
                continue;
            }
        }
        if (!hit) continue;   // :268
        hit_any = true;       // :270
        tmax = rec.t;         // :271
    }
    if (hit_any) st.counters.hits++;
    return hit_any;
}

// ---------------------------------------------------------------------------
// src/camera.rs:311-320,368-380 — primary ray
// ---------------------------------------------------------------------------
template <class R>
inline Ray get_ray(const RaytracingCamera& c, R& rng, size_t x, size_t y) {
    Vec3 pixel_centre = c.pixel00_loc + ((float)x * c.pixel_delta_u) + ((float)y * c.pixel_delta_v);
    float px = -0.5f + rng.camera();
    float py = -0.5f + rng.camera();
    Vec3 pixel_sample = pixel_centre + (px * c.pixel_delta_u + py * c.pixel_delta_v);
    Vec3 ray_origin;
    if (c.defocus_angle <= 0.0f) ray_origin = c.centre;
    else {
        Vec3 p = random_in_unit_disk([&]() { return rng.camera(); });
        ray_origin = c.centre + (p.x * c.defocus_disk_u) + (p.y * c.defocus_disk_v);
    }
    Vec3 ray_direction = pixel_sample - ray_origin;
    float ray_time = rng.camera();
    return Ray{ray_origin, ray_direction, ray_time};
}

// src/camera.rs:324-365 — iterative path
template <class R>
inline Vec3 ray_colour(const RaytracingCamera& c, R& rng, Ray ray, const Scene& scene, Stacks& st) {
    size_t depth = c.max_depth;
    Vec3 multiplier(1.0f, 1.0f, 1.0f);
    HitRecord rec;
    uint32_t bounce = 0;
    for (;;) {
        if (depth == 0) return Vec3(0, 0, 0);
        rng.begin_bounce(bounce);
        bool hit = hit_anything(ray, rng, rec, scene.entry, st);
        if (!hit) return multiplier * c.background_colour;
        const Material& material = scene.mmap.map[rec.material];
        Vec3 emission = material.emitted(rec.u, rec.v, rec.point);
        Ray scattered; Vec3 attenuation;
        if (!material.scatter(rng, ray, rec, scattered, attenuation)) return multiplier * emission;
        ray = scattered;
        depth -= 1;
        multiplier = multiplier * attenuation + emission;
        ++bounce;
    }
}

enum RngMode { RNG_XOSHIRO = 0, RNG_COUNTER = 1 };

// src/camera.rs:269-306 — one call of the reference = one sample index; this
// loops over [first_sample, first_sample + n_samples).  Rows are handed to
// `threads` workers through an atomic counter (the reference uses rayon's
// par_bridge over rows, :282-284).  acc is W*H*4 floats, lane 3 untouched.
inline void render(const Scene& scene, RngMode mode, uint64_t seed, uint32_t first_sample, uint32_t n_samples,
                   float* acc, int threads, Counters* counters_out) {
    const RaytracingCamera& c = scene.rt_cam;
    const size_t W = c.width, H = c.height;
    if (threads < 1) threads = 1;
    std::vector<Counters> per_thread((size_t)threads);
    for (uint32_t s = first_sample; s < first_sample + n_samples; ++s) {
        std::atomic<size_t> next_row{0};
        auto worker = [&](int tid) {
            Stacks st;
            for (;;) {
                size_t y = next_row.fetch_add(1);
                if (y >= H) break;
                float* row = acc + 4 * y * W;
                if (mode == RNG_XOSHIRO) {
                    RngXoshiro rng{Seed{{(uint64_t)y, (uint64_t)s, (uint64_t)(W * H), (uint64_t)y}}};  // :287
                    for (size_t x = 0; x < W; ++x) {
                        Ray ray = get_ray(c, rng, x, y);
                        Vec3 col = ray_colour(c, rng, ray, scene, st);
                        row[4 * x + 0] += col.x; row[4 * x + 1] += col.y; row[4 * x + 2] += col.z;
                    }
                } else {
                    RngCounter rng; rng.seed = seed;
                    for (size_t x = 0; x < W; ++x) {
                        rng.begin_path((uint32_t)(y * W + x), s);
                        Ray ray = get_ray(c, rng, x, y);
                        Vec3 col = ray_colour(c, rng, ray, scene, st);
                        row[4 * x + 0] += col.x; row[4 * x + 1] += col.y; row[4 * x + 2] += col.z;
                    }
                }
                st.counters.samples += W;
            }
            per_thread[(size_t)tid].add(st.counters);
        };
        if (threads == 1) worker(0);
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t);
            for (auto& th : pool) th.join();
        }
    }
    if (counters_out) for (auto& c2 : per_thread) counters_out->add(c2);
}

}  // namespace orc
