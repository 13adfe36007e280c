// oracle_capi.cpp — C entry points of the CPU oracle, for ctypes.
// TEST INFRASTRUCTURE ONLY: see the header of oracle_core.hpp.  The plain-data
// structs of include/rt_abi.h are reused for rays, hits, cameras and counters so
// that tests can hand the same buffers to the oracle and to the GPU library.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "../include/rt_abi.h"
#include "oracle_core.hpp"
#include "oracle_scenes.hpp"

using namespace orc;

namespace {

struct Handle {
    Scene scene;
    std::vector<Hittable> handles;  // builder API: hittables by value, like Rust moves
    Image earth;
};

Vec3 v3(const float* p) { return Vec3(p[0], p[1], p[2]); }
void st3(float* d, Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

void load_rgb8(Image& im, const uint8_t* data, uint32_t w, uint32_t h) {
    im.width = w; im.height = h;
    im.rgb.resize((size_t)w * h * 3);
    // image::DynamicImage::into_rgb32f on 8-bit data: value / 255 in f32
    for (size_t i = 0; i < im.rgb.size(); ++i) im.rgb[i] = (float)data[i] / 255.0f;
}

struct Leaf {
    uint32_t kind, prim_id, material, medium;  // medium: index of the enclosing medium boundary or 0xFFFFFFFF
    float data[12];
    float sin_theta, cos_theta, offset[3];
    uint32_t has_transform;
};

struct Xf { bool any = false; float s = 0, c = 1; Vec3 off; };

// compose: child-space point = R_child * (parent-space point - off_child)
Xf compose_move(const Xf& outer, Vec3 offset) {
    // outer: p1 = R(p - off); then move: p2 = p1 - offset = R(p - off - R^-1 offset)
    Xf r = outer; r.any = true;
    Vec3 inv(outer.c * offset.x + outer.s * offset.z, offset.y, -outer.s * offset.x + outer.c * offset.z);
    r.off = outer.off + inv;
    return r;
}
Xf compose_rot(const Xf& outer, float s, float c) {
    // p2 = R2 * R(p - off): angles add
    Xf r = outer; r.any = true;
    r.s = s * outer.c + c * outer.s;
    r.c = c * outer.c - s * outer.s;
    return r;
}

void walk(const Hittable* h, const Xf& xf, uint32_t medium, std::vector<Leaf>& out) {
    Leaf l{};
    l.medium = medium;
    l.has_transform = xf.any ? 1u : 0u;
    l.sin_theta = xf.s; l.cos_theta = xf.c; st3(l.offset, xf.off);
    switch (h->kind) {
        case Hittable::SPHERE:
            l.kind = RT_PRIM_SPHERE; l.prim_id = h->sphere.prim_id; l.material = h->sphere.material;
            st3(l.data, h->sphere.centre); l.data[3] = h->sphere.radius; out.push_back(l); break;
        case Hittable::MOVING_SPHERE:
            l.kind = RT_PRIM_MOVING_SPHERE; l.prim_id = h->moving.prim_id; l.material = h->moving.material;
            st3(l.data, h->moving.centre.origin); l.data[3] = h->moving.radius; st3(l.data + 4, h->moving.centre.direction);
            out.push_back(l); break;
        case Hittable::QUAD:
            l.kind = RT_PRIM_QUAD; l.prim_id = h->quad.prim_id; l.material = h->quad.material;
            st3(l.data, h->quad.q); st3(l.data + 3, h->quad.u); st3(l.data + 6, h->quad.v); out.push_back(l); break;
        case Hittable::CONSTANT_MEDIUM:
            l.kind = RT_PRIM_MEDIUM; l.prim_id = h->medium.prim_id; l.material = h->medium.phase_function;
            l.data[0] = h->medium.neg_inv_density; l.data[1] = (float)h->medium.medium_index; out.push_back(l);
            walk(h->medium.boundary, xf, h->medium.medium_index, out);
            break;
        case Hittable::BVH:
            walk(h->bvh.left, xf, medium, out);
            if (h->bvh.right != h->bvh.left) walk(h->bvh.right, xf, medium, out);
            break;
        case Hittable::MOVE: walk(h->move.obj, compose_move(xf, h->move.offset), medium, out); break;
        case Hittable::ROTATE_Y: walk(h->rot.obj, compose_rot(xf, h->rot.sin, h->rot.cos), medium, out); break;
    }
}

void fill_hit(rt_hit& o, bool hit, const HitRecord& rec) {
    std::memset(&o, 0, sizeof(o));
    if (!hit) return;
    o.hit = 1; o.t = rec.t; st3(o.point, rec.point); st3(o.normal, rec.normal); o.u = rec.u; o.v = rec.v;
    o.prim_id = rec.prim_id; o.material = rec.material; o.front_face = rec.front_face ? 1u : 0u;
}

}  // namespace

extern "C" {

void* orc_scene_new() { return new Handle(); }
void orc_scene_free(void* p) { delete (Handle*)p; }

void* orc_scene_builtin(int id, uint32_t width, uint32_t height, uint32_t max_depth, const uint8_t* earth_rgb8,
                        uint32_t earth_w, uint32_t earth_h, uint64_t stress_count) {
    Handle* h = new Handle();
    BuildOptions o;
    o.width = width; o.height = height; o.max_depth = max_depth; o.stress_count = stress_count;
    if (earth_rgb8) { load_rgb8(h->earth, earth_rgb8, earth_w, earth_h); o.earth = &h->earth; }
    if (!build_builtin(h->scene, id, o)) { delete h; return nullptr; }
    return h;
}

// ---- builder API (mirrors the reference's constructors) ---------------------
uint32_t orc_tex_solid(void* p, float r, float g, float b) { return ((Handle*)p)->scene.tex_colour(Vec3(r, g, b))->id; }
uint32_t orc_tex_checker(void* p, float scale, uint32_t even, uint32_t odd) {
    Scene& s = ((Handle*)p)->scene;
    return s.tex_checker(scale, &s.textures[even], &s.textures[odd])->id;
}
uint32_t orc_tex_image_rgb8(void* p, const uint8_t* data, uint32_t w, uint32_t h) {
    Scene& s = ((Handle*)p)->scene;
    s.images.emplace_back(); load_rgb8(s.images.back(), data, w, h);
    return s.tex_image(&s.images.back())->id;
}
uint32_t orc_tex_image_rgb32f(void* p, const float* data, uint32_t w, uint32_t h) {
    Scene& s = ((Handle*)p)->scene;
    s.images.emplace_back();
    Image& im = s.images.back(); im.width = w; im.height = h; im.rgb.assign(data, data + (size_t)w * h * 3);
    return s.tex_image(&im)->id;
}
uint32_t orc_perlin(void* p, const uint64_t* seed4, uint32_t point_count) {
    Scene& s = ((Handle*)p)->scene;
    Seed seed{{seed4[0], seed4[1], seed4[2], seed4[3]}};
    s.perlins.push_back(PerlinNoise::make(seed, point_count));
    return (uint32_t)s.perlins.size() - 1;
}
uint32_t orc_tex_noise(void* p, uint32_t perlin, float scale) {
    Scene& s = ((Handle*)p)->scene;
    return s.tex_noise(&s.perlins[perlin], scale)->id;
}
uint32_t orc_material(void* p, uint32_t kind, uint32_t tex, float param) {
    Scene& s = ((Handle*)p)->scene;
    return s.push_material((Material::Kind)kind, &s.textures[tex], param);
}
static uint32_t push_handle(Handle* h, const Hittable& x) { h->handles.push_back(x); return (uint32_t)h->handles.size() - 1; }
uint32_t orc_sphere(void* p, const float* c, float r, uint32_t mat) { Handle* h = (Handle*)p; return push_handle(h, h->scene.sphere(v3(c), r, mat)); }
uint32_t orc_moving_sphere(void* p, const float* c1, const float* c2, float r, uint32_t mat) {
    Handle* h = (Handle*)p; return push_handle(h, h->scene.moving_sphere(v3(c1), v3(c2), r, mat));
}
uint32_t orc_quad(void* p, const float* q, const float* u, const float* v, uint32_t mat) {
    Handle* h = (Handle*)p; return push_handle(h, h->scene.quad(v3(q), v3(u), v3(v), mat));
}
uint32_t orc_box(void* p, const float* a, const float* b, uint32_t mat) {
    Handle* h = (Handle*)p; return push_handle(h, h->scene.box_of_quads(v3(a), v3(b), mat));
}
uint32_t orc_bvh(void* p, const uint32_t* hs, uint32_t n) {
    Handle* h = (Handle*)p;
    std::vector<Hittable> list;
    for (uint32_t i = 0; i < n; ++i) list.push_back(h->handles[hs[i]]);
    return push_handle(h, h->scene.bvh(list));
}
uint32_t orc_move(void* p, uint32_t x, const float* off) { Handle* h = (Handle*)p; return push_handle(h, h->scene.move_by(h->handles[x], v3(off))); }
uint32_t orc_rotate_y(void* p, uint32_t x, float degrees) { Handle* h = (Handle*)p; return push_handle(h, h->scene.rotate_y_by(h->handles[x], degrees)); }
uint32_t orc_medium(void* p, uint32_t boundary, float density, uint32_t tex) {
    Handle* h = (Handle*)p;
    const Hittable* b = h->scene.alloc(h->handles[boundary]);
    return push_handle(h, h->scene.constant_medium(b, density, &h->scene.textures[tex]));
}
void orc_set_world(void* p, uint32_t x) { Handle* h = (Handle*)p; h->scene.entry = h->scene.alloc(h->handles[x]); }
void orc_camera(void* p, const float* pos, uint32_t w, uint32_t hgt, uint32_t depth, float vfov, const float* vup, float defocus,
                float focus, const float* bg, float pitch, float yaw) {
    Scene& s = ((Handle*)p)->scene;
    s.camera_new(v3(pos), w, hgt, depth, vfov, v3(vup), defocus, focus, v3(bg));
    s.change_pitch_yaw_by(pitch, yaw);
}

// ---- queries ------------------------------------------------------------------
void orc_camera_desc(void* p, rt_camera_desc* out) {
    const RaytracingCamera& c = ((Handle*)p)->scene.rt_cam;
    std::memset(out, 0, sizeof(*out));
    out->width = (uint32_t)c.width; out->height = (uint32_t)c.height; out->max_depth = (uint32_t)c.max_depth;
    out->defocus_angle = c.defocus_angle;
    st3(out->centre, c.centre); st3(out->pixel00_loc, c.pixel00_loc); st3(out->pixel_delta_u, c.pixel_delta_u);
    st3(out->pixel_delta_v, c.pixel_delta_v); st3(out->defocus_disk_u, c.defocus_disk_u); st3(out->defocus_disk_v, c.defocus_disk_v);
    st3(out->background_colour, c.background_colour);
}
uint32_t orc_material_count(void* p) { return (uint32_t)((Handle*)p)->scene.mmap.map.size(); }
void orc_material_get(void* p, uint32_t i, uint32_t* kind, float* param, uint32_t* tex_kind, float* colour3) {
    const Material& m = ((Handle*)p)->scene.mmap.map[i];
    *kind = (uint32_t)m.kind; *param = m.param; *tex_kind = (uint32_t)m.texture->kind; st3(colour3, m.texture->colour);
}
uint32_t orc_bvh_assert_failures(void* p) { return ((Handle*)p)->scene.bvh_assert_failures; }

// Leaves in tree order; call with out == NULL to get the count.
uint32_t orc_export_leaves(void* p, void* out, uint32_t cap) {
    Handle* h = (Handle*)p;
    std::vector<Leaf> leaves;
    walk(h->scene.entry, Xf{}, 0xFFFFFFFFu, leaves);
    if (out) std::memcpy(out, leaves.data(), sizeof(Leaf) * std::min<size_t>(cap, leaves.size()));
    return (uint32_t)leaves.size();
}
uint32_t orc_perlin_count(void* p) { return (uint32_t)((Handle*)p)->scene.perlins.size(); }
uint32_t orc_perlin_export(void* p, uint32_t idx, float* vecs, uint32_t* px, uint32_t* py, uint32_t* pz) {
    const PerlinNoise& pn = ((Handle*)p)->scene.perlins[idx];
    uint32_t n = (uint32_t)pn.rand_floats.size();
    if (vecs) for (uint32_t i = 0; i < n; ++i) { st3(vecs + 3 * i, pn.rand_floats[i]); px[i] = (uint32_t)pn.perm_x[i]; py[i] = (uint32_t)pn.perm_y[i]; pz[i] = (uint32_t)pn.perm_z[i]; }
    return n;
}

// ---- the render path ------------------------------------------------------------
void orc_render(void* p, int mode, uint64_t seed, uint32_t first_sample, uint32_t n_samples, float* acc, int threads,
                rt_counters* counters) {
    Counters c;
    render(((Handle*)p)->scene, (RngMode)mode, seed, first_sample, n_samples, acc, threads, &c);
    if (counters) {
        counters->samples += c.samples; counters->segments += c.segments; counters->node_visits += c.node_visits;
        counters->sphere_tests += c.sphere_tests; counters->quad_tests += c.quad_tests; counters->medium_tests += c.medium_tests;
        counters->hits += c.hits;
    }
}

void orc_intersect(void* p, const rt_ray* rays, size_t n, uint64_t seed, rt_hit* out) {
    const Scene& s = ((Handle*)p)->scene;
    Stacks st;
    for (size_t i = 0; i < n; ++i) {
        const rt_ray& r = rays[i];
        RngCounter rng; rng.seed = seed;
        rng.begin_path(r.rng_pixel, r.rng_sample);
        rng.begin_bounce(r.rng_bounce);
        Ray ray{v3(r.origin), v3(r.direction), r.time};
        HitRecord rec;
        bool hit = hit_anything(ray, rng, rec, s.entry, st, r.tmin, r.tmax);
        fill_hit(out[i], hit, rec);
    }
}

// Material::emitted + Material::scatter on given hit records, counter RNG keyed per ray.
// out_scattered[i].tmin is set to 1 when scatter returned Some, else 0.
void orc_shade(void* p, const rt_ray* rays, const rt_hit* hits, size_t n, uint64_t seed, rt_ray* out_scattered,
               float* out_attenuation3, float* out_emitted3) {
    const Scene& s = ((Handle*)p)->scene;
    for (size_t i = 0; i < n; ++i) {
        const rt_ray& r = rays[i];
        const rt_hit& h = hits[i];
        RngCounter rng; rng.seed = seed;
        rng.begin_path(r.rng_pixel, r.rng_sample);
        rng.begin_bounce(r.rng_bounce);
        Ray ray{v3(r.origin), v3(r.direction), r.time};
        HitRecord rec;
        rec.t = h.t; rec.point = v3(h.point); rec.normal = v3(h.normal); rec.u = h.u; rec.v = h.v;
        rec.material = h.material; rec.front_face = h.front_face != 0;
        const Material& m = s.mmap.map[rec.material];
        st3(out_emitted3 + 3 * i, m.emitted(rec.u, rec.v, rec.point));
        Ray sc; Vec3 att;
        bool ok = m.scatter(rng, ray, rec, sc, att);
        std::memset(&out_scattered[i], 0, sizeof(rt_ray));
        out_attenuation3[3 * i] = out_attenuation3[3 * i + 1] = out_attenuation3[3 * i + 2] = 0.0f;
        if (ok) {
            st3(out_scattered[i].origin, sc.origin); st3(out_scattered[i].direction, sc.direction);
            out_scattered[i].time = sc.time; out_scattered[i].tmin = 1.0f;
            st3(out_attenuation3 + 3 * i, att);
        }
    }
}

// Texture::value of material `mat`'s texture
void orc_texture_value(void* p, uint32_t mat, size_t n, const float* uv2, const float* p3, float* out3) {
    const Scene& s = ((Handle*)p)->scene;
    const Texture* t = s.mmap.map[mat].texture;
    for (size_t i = 0; i < n; ++i) st3(out3 + 3 * i, t->value(uv2[2 * i], uv2[2 * i + 1], v3(p3 + 3 * i)));
}
void orc_perlin_eval(void* p, uint32_t perlin, size_t n, const float* p3, uint32_t depth, float* out_noise, float* out_turb) {
    const PerlinNoise& pn = ((Handle*)p)->scene.perlins[perlin];
    for (size_t i = 0; i < n; ++i) { out_noise[i] = pn.noise(v3(p3 + 3 * i)); out_turb[i] = pn.turbulance(v3(p3 + 3 * i), depth); }
}

// primary rays (get_ray) with the counter RNG, for raygen parity
void orc_primary_rays(void* p, uint64_t seed, uint32_t sample, const uint32_t* pixels, size_t n, rt_ray* out) {
    const Scene& s = ((Handle*)p)->scene;
    const RaytracingCamera& c = s.rt_cam;
    for (size_t i = 0; i < n; ++i) {
        RngCounter rng; rng.seed = seed;
        rng.begin_path(pixels[i], sample);
        Ray r = get_ray(c, rng, pixels[i] % c.width, pixels[i] / c.width);
        std::memset(&out[i], 0, sizeof(rt_ray));
        st3(out[i].origin, r.origin); st3(out[i].direction, r.direction); out[i].time = r.time;
        out[i].tmin = 0.001f; out[i].tmax = INF; out[i].rng_pixel = pixels[i]; out[i].rng_sample = sample;
    }
}

// ---- known-answer helpers ---------------------------------------------------------
void orc_seed_next(uint64_t* state4, uint32_t n, uint64_t* out) {
    Seed s{{state4[0], state4[1], state4[2], state4[3]}};
    for (uint32_t i = 0; i < n; ++i) out[i] = s.next();
    std::memcpy(state4, s.s, sizeof(s.s));
}
void orc_seed_next_f32(uint64_t* state4, uint32_t n, float* out) {
    Seed s{{state4[0], state4[1], state4[2], state4[3]}};
    for (uint32_t i = 0; i < n; ++i) out[i] = s.next_f32();
    std::memcpy(state4, s.s, sizeof(s.s));
}
void orc_counter_draws(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stream, uint32_t n, uint32_t* out_u32,
                       float* out_f32) {
    CounterStream cs;
    cs.init(CounterStream::path_key(seed, pixel, sample), bounce, stream);
    for (uint32_t i = 0; i < n; ++i) {
        CounterStream copy = cs;
        out_u32[i] = cs.next_u32();
        out_f32[i] = copy.next_f32();
    }
}

// Camera::hdr_render / realtime_render closures (src/camera.rs:95-100,115-118)
void orc_finalize(const float* acc, uint64_t n_pixels, uint32_t total_samples, int mode, float exposure, void* out) {
    float inv = 1.0f / (float)total_samples;  // src/camera.rs:279
    for (uint64_t i = 0; i < n_pixels; ++i) {
        Vec3 c = inv * Vec3(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2]);
        if (mode == RT_FINALIZE_MEAN) {
            float* o = (float*)out + 4 * i; o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = 0.0f;
        } else if (mode == RT_FINALIZE_SQRT) {
            float* o = (float*)out + 4 * i; o[0] = std::sqrt(c.x); o[1] = std::sqrt(c.y); o[2] = std::sqrt(c.z); o[3] = 0.0f;
        } else {
            Vec3 e = exposure * -c;
            Vec3 mapped(1.0f - std::exp(e.x), 1.0f - std::exp(e.y), 1.0f - std::exp(e.z));
            ((uint32_t*)out)[i] = to_rgba(mapped);
        }
    }
}

}  // extern "C"
