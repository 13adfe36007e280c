// capi.cpp — C entry points over the host API of raytracing.hpp, for callers
// without a C++ toolchain (the Python package binds these with ctypes).  Every
// function returns 0 or -1; rth_last_error() describes the last failure.
#include <cstring>
#include <string>

#include "raytracing.hpp"

using namespace raytracing;

namespace {

thread_local std::string g_err;

struct Handle {
    Arena arena;
    std::unique_ptr<Camera> camera;
    const Rgb32FImage* earth = nullptr;
    // builder state
    MaterialMap mmap;
    std::vector<Texture> textures;
    std::vector<PerlinNoise> perlins;
    std::vector<Hittable> handles;
    const Hittable* entry = nullptr;
    int device = 0;
};

Vec3 v3(const float* p) { return Vec3(p[0], p[1], p[2]); }

template <class F>
int guard(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

struct Leaf {  // same layout as the leaf records the CPU oracle exports
    uint32_t kind, prim_id, material, medium;
    float data[12];
    float sin_theta, cos_theta, offset[3];
    uint32_t has_transform;
};

}  // namespace

extern "C" {

const char* rth_last_error() { return g_err.c_str(); }

void* rth_scene_new(int device) {
    Handle* h = new Handle();
    h->device = device;
    return h;
}
void rth_scene_free(void* p) { delete (Handle*)p; }

void* rth_scene_builtin(int id, uint32_t width, uint32_t height, uint32_t max_depth, const uint8_t* earth_rgb8, uint32_t earth_w,
                        uint32_t earth_h, uint64_t stress_count, int device) {
    Handle* h = new Handle();
    h->device = device;
    int rc = guard([&] {
        SceneOptions o;
        o.width = width; o.height = height; o.max_depth = max_depth; o.stress_count = stress_count; o.device = device;
        if (earth_rgb8) {
            Rgb32FImage im;
            im.width = earth_w; im.height = earth_h;
            im.rgb8.assign(earth_rgb8, earth_rgb8 + (size_t)earth_w * earth_h * 3);
            h->earth = h->arena.alloc_new(std::move(im));
            o.earth = h->earth;
        }
        h->camera = build_scene(h->arena, id, o);
    });
    if (rc) { delete h; return nullptr; }
    return h;
}

// ---- builder API (same shape as the reference's constructors) --------------------
uint32_t rth_tex_solid(void* p, float r, float g, float b) {
    Handle* h = (Handle*)p; h->textures.push_back(Texture::colour(Colour(r, g, b))); return (uint32_t)h->textures.size() - 1;
}
uint32_t rth_tex_checker(void* p, float scale, uint32_t even, uint32_t odd) {
    Handle* h = (Handle*)p;
    const Texture* e = h->arena.alloc_new(h->textures[even]);
    const Texture* o = h->arena.alloc_new(h->textures[odd]);
    h->textures.push_back(Texture::checkerboard(scale, e, o));
    return (uint32_t)h->textures.size() - 1;
}
uint32_t rth_tex_image_rgb8(void* p, const uint8_t* data, uint32_t w, uint32_t hgt) {
    Handle* h = (Handle*)p;
    Rgb32FImage im; im.width = w; im.height = hgt; im.rgb8.assign(data, data + (size_t)w * hgt * 3);
    h->textures.push_back(Texture::image(h->arena.alloc_new(std::move(im))));
    return (uint32_t)h->textures.size() - 1;
}
uint32_t rth_tex_image_rgb32f(void* p, const float* data, uint32_t w, uint32_t hgt) {
    Handle* h = (Handle*)p;
    Rgb32FImage im; im.width = w; im.height = hgt; im.rgb32f.assign(data, data + (size_t)w * hgt * 3);
    h->textures.push_back(Texture::image(h->arena.alloc_new(std::move(im))));
    return (uint32_t)h->textures.size() - 1;
}
uint32_t rth_perlin(void* p, const uint64_t* seed4, uint32_t point_count) {
    Handle* h = (Handle*)p;
    Seed seed{{seed4[0], seed4[1], seed4[2], seed4[3]}};
    h->perlins.push_back(PerlinNoise::make(h->arena, seed, point_count));
    return (uint32_t)h->perlins.size() - 1;
}
uint32_t rth_tex_noise(void* p, uint32_t perlin, float scale) {
    Handle* h = (Handle*)p; h->textures.push_back(Texture::noise(h->perlins[perlin], scale)); return (uint32_t)h->textures.size() - 1;
}
uint32_t rth_material(void* p, uint32_t kind, uint32_t tex, float param) {
    Handle* h = (Handle*)p;
    Material m;
    m.kind = (Material::Kind)kind; m.texture = h->textures[tex]; m.param = param;
    return h->mmap.push(m).v;
}
static uint32_t push_handle(Handle* h, Hittable x) { h->handles.push_back(std::move(x)); return (uint32_t)h->handles.size() - 1; }
uint32_t rth_sphere(void* p, const float* c, float r, uint32_t mat) {
    return push_handle((Handle*)p, Hittable::sphere(Sphere::make(v3(c), r, MaterialId{mat})));
}
uint32_t rth_moving_sphere(void* p, const float* c1, const float* c2, float r, uint32_t mat) {
    return push_handle((Handle*)p, Hittable::moving_sphere(MovingSphere::make(v3(c1), v3(c2), r, MaterialId{mat})));
}
uint32_t rth_quad(void* p, const float* q, const float* u, const float* v, uint32_t mat) {
    return push_handle((Handle*)p, Hittable::quad(Quad::make(v3(q), v3(u), v3(v), MaterialId{mat})));
}
uint32_t rth_box(void* p, const float* a, const float* b, uint32_t mat) {
    Handle* h = (Handle*)p; return push_handle(h, Hittable::box_of_quads(h->arena, v3(a), v3(b), MaterialId{mat}));
}
uint32_t rth_bvh(void* p, const uint32_t* hs, uint32_t n) {
    Handle* h = (Handle*)p;
    std::vector<Hittable> list;
    for (uint32_t i = 0; i < n; ++i) list.push_back(h->handles[hs[i]]);
    return push_handle(h, Hittable::bvh(h->arena, list));
}
uint32_t rth_move(void* p, uint32_t x, const float* off) { Handle* h = (Handle*)p; return push_handle(h, h->handles[x].move_by(h->arena, v3(off))); }
uint32_t rth_rotate_y(void* p, uint32_t x, float degrees) { Handle* h = (Handle*)p; return push_handle(h, h->handles[x].rotate_y_by(h->arena, degrees)); }
uint32_t rth_medium(void* p, uint32_t boundary, float density, uint32_t tex) {
    Handle* h = (Handle*)p;
    const Hittable* b = h->arena.alloc_new(h->handles[boundary]);
    return push_handle(h, Hittable::constant_medium(ConstantMedium::make(h->mmap, b, density, h->textures[tex])));
}
void rth_set_world(void* p, uint32_t x) { Handle* h = (Handle*)p; h->entry = h->arena.alloc_new(h->handles[x]); }
// Camera::new + change_pitch_yaw_by + set_world with what the builder calls above collected
int rth_camera(void* p, const float* pos, uint32_t w, uint32_t hgt, uint32_t depth, float vfov, const float* vup, float defocus, float focus,
               const float* bg, float pitch, float yaw) {
    Handle* h = (Handle*)p;
    return guard([&] {
        h->camera = std::make_unique<Camera>(h->arena, v3(pos), Vec3(1.0f, 0.0f, 0.0f), std::make_pair((size_t)w, (size_t)hgt), 1.0f,
                                             (size_t)depth, vfov, v3(vup), defocus, focus, v3(bg), h->device);
        h->camera->change_pitch_yaw_by(pitch, yaw);
        h->camera->set_world(World::make(h->entry, h->mmap));
    });
}

// ---- camera operations ---------------------------------------------------------------
static Camera* cam(void* p) {
    Handle* h = (Handle*)p;
    if (!h->camera) throw RenderError("no camera: call rth_camera first");
    return h->camera.get();
}
int rth_set_seed(void* p, uint64_t seed) { return guard([&] { cam(p)->set_seed(seed); }); }
int rth_render_samples(void* p, uint32_t n) { return guard([&] { cam(p)->render_samples(n); }); }
int rth_empty_render(void* p) { return guard([&] { cam(p)->empty_render(); }); }
int rth_hdr_render(void* p, float* out_rgb) {
    return guard([&] {
        std::vector<Vec3> img = cam(p)->hdr_render();
        for (size_t i = 0; i < img.size(); ++i) { out_rgb[3 * i] = img[i].x; out_rgb[3 * i + 1] = img[i].y; out_rgb[3 * i + 2] = img[i].z; }
    });
}
int rth_realtime_render(void* p, uint32_t* out) {
    return guard([&] {
        const Camera::Words img = cam(p)->realtime_render();
        std::memcpy(out, img.data(), img.size() * sizeof(uint32_t));
    });
}
// The same without the copy: *words points into the camera's own (page-locked) frame, valid until its next render call.
int rth_realtime_render_view(void* p, const uint32_t** words, uint64_t* n_words) {
    return guard([&] {
        const Camera::Words img = cam(p)->realtime_render();
        *words = img.data();
        *n_words = img.size();
    });
}
int rth_download_accumulator(void* p, float* out) {
    return guard([&] {
        std::vector<float> acc = cam(p)->download_accumulator();
        std::memcpy(out, acc.data(), acc.size() * sizeof(float));
    });
}
// out_rgba may be NULL; path may be NULL or empty (no file)
int rth_render_to_image(void* p, uint64_t sample_count, const char* path, float* out_rgba) {
    return guard([&] {
        std::vector<float> img = cam(p)->render_to_image((size_t)sample_count, path ? path : "");
        if (out_rgba) std::memcpy(out_rgba, img.data(), img.size() * sizeof(float));
    });
}
int rth_save_checkpoint(void* p, const char* path) { return guard([&] { cam(p)->save_checkpoint(path); }); }
int rth_load_checkpoint(void* p, const char* path) { return guard([&] { cam(p)->load_checkpoint(path); }); }
int rth_write_exr(const char* path, uint32_t w, uint32_t h, const float* rgba) { return guard([&] { write_exr(path, w, h, rgba); }); }
int rth_write_pfm(const char* path, uint32_t w, uint32_t h, const float* rgba) { return guard([&] { write_pfm(path, w, h, rgba); }); }
int rth_change_pitch_yaw_by(void* p, float dp, float dy) { return guard([&] { cam(p)->change_pitch_yaw_by(dp, dy); }); }
int rth_move_by(void* p, const float* step) { return guard([&] { cam(p)->move_by(v3(step)); }); }
int rth_set_exposure(void* p, float e) { return guard([&] { cam(p)->set_exposure(e); }); }
uint64_t rth_samples(void* p) { Handle* h = (Handle*)p; return h->camera ? h->camera->samples() : 0; }
int rth_camera_desc(void* p, rt_camera_desc* out) { return guard([&] { *out = cam(p)->raytracing_camera_desc(); }); }
// NULL on failure (see rth_last_error)
rt_ctx* rth_ctx(void* p) { rt_ctx* r = nullptr; guard([&] { r = cam(p)->context(); }); return r; }
rt_scene* rth_scene(void* p) { rt_scene* r = nullptr; guard([&] { r = cam(p)->scene(); }); return r; }
float* rth_device_accumulator(void* p) { float* r = nullptr; guard([&] { r = cam(p)->device_accumulator(); }); return r; }

// ---- inspection of the flattened scene -------------------------------------------------
// out[0..10] = spheres, moving spheres, quads, media, world refs, boundary refs, transforms,
// materials, textures, images, perlins
int rth_flat_counts(void* p, uint32_t* out) {
    return guard([&] {
        rt_scene_desc d = cam(p)->flat_scene().desc();
        uint32_t v[11] = {d.n_spheres, d.n_moving_spheres, d.n_quads, d.n_media, d.n_world, d.n_boundary,
                          d.n_transforms, d.n_materials, d.n_textures, d.n_images, d.n_perlins};
        std::memcpy(out, v, sizeof(v));
    });
}
// World leaves, then each medium's boundary leaves; out == NULL returns the count.
uint32_t rth_export_leaves(void* p, void* out, uint32_t cap) {
    Handle* h = (Handle*)p;
    if (!h->camera) return 0;
    const FlatScene& f = h->camera->flat_scene();
    std::vector<Leaf> leaves;
    auto emit = [&](const rt_prim_ref& r, uint32_t medium) {
        Leaf l{};
        l.kind = r.kind; l.medium = medium;
        uint32_t xf = RT_NO_TRANSFORM;
        switch (r.kind) {
            case RT_PRIM_SPHERE: {
                const rt_sphere& s = f.spheres[r.index];
                l.prim_id = s.prim_id; l.material = s.material; xf = s.transform;
                std::memcpy(l.data, s.centre, 12); l.data[3] = s.radius; break;
            }
            case RT_PRIM_MOVING_SPHERE: {
                const rt_moving_sphere& s = f.moving_spheres[r.index];
                l.prim_id = s.prim_id; l.material = s.material; xf = s.transform;
                std::memcpy(l.data, s.centre0, 12); l.data[3] = s.radius; std::memcpy(l.data + 4, s.dcentre, 12); break;
            }
            case RT_PRIM_QUAD: {
                const rt_quad& q = f.quads[r.index];
                l.prim_id = q.prim_id; l.material = q.material; xf = q.transform;
                std::memcpy(l.data, q.q, 12); std::memcpy(l.data + 3, q.u, 12); std::memcpy(l.data + 6, q.v, 12); break;
            }
            default: {
                const rt_medium& m = f.media[r.index];
                l.prim_id = m.prim_id; l.material = m.phase_material; l.data[0] = m.neg_inv_density; l.data[1] = (float)r.index; break;
            }
        }
        l.cos_theta = 1.0f;
        if (xf != RT_NO_TRANSFORM) {
            const rt_transform& t = f.transforms[xf];
            l.has_transform = 1; l.sin_theta = t.sin_theta; l.cos_theta = t.cos_theta; std::memcpy(l.offset, t.offset, 12);
        }
        leaves.push_back(l);
    };
    for (const rt_prim_ref& r : f.world) {
        emit(r, 0xFFFFFFFFu);
        if (r.kind == RT_PRIM_MEDIUM) {
            const rt_medium& m = f.media[r.index];
            for (uint32_t k = 0; k < m.n_boundary; ++k) emit(f.boundary[m.first_boundary + k], r.index);
        }
    }
    if (out) std::memcpy(out, leaves.data(), sizeof(Leaf) * std::min<size_t>(cap, leaves.size()));
    return (uint32_t)leaves.size();
}
uint32_t rth_material_count(void* p) { Handle* h = (Handle*)p; return h->camera ? (uint32_t)h->camera->flat_scene().materials.size() : 0; }
int rth_material_get(void* p, uint32_t i, uint32_t* kind, float* param, uint32_t* tex_kind, float* colour3) {
    return guard([&] {
        const FlatScene& f = cam(p)->flat_scene();
        const rt_material& m = f.materials.at(i);
        const rt_texture& t = f.textures.at(m.texture);
        *kind = m.kind; *param = m.param; *tex_kind = t.kind; std::memcpy(colour3, t.colour, 12);
    });
}
uint32_t rth_perlin_count(void* p) { Handle* h = (Handle*)p; return h->camera ? (uint32_t)h->camera->flat_scene().perlins.size() : 0; }
uint32_t rth_perlin_export(void* p, uint32_t idx, float* vecs, uint32_t* px, uint32_t* py, uint32_t* pz) {
    Handle* h = (Handle*)p;
    if (!h->camera) return 0;
    const rt_perlin& pn = h->camera->flat_scene().perlins[idx];
    if (vecs) {
        std::memcpy(vecs, pn.rand_vecs, sizeof(float) * 3 * pn.point_count);
        std::memcpy(px, pn.perm_x, 4 * pn.point_count); std::memcpy(py, pn.perm_y, 4 * pn.point_count); std::memcpy(pz, pn.perm_z, 4 * pn.point_count);
    }
    return pn.point_count;
}
// The flattened description itself, for callers that drive rt_scene_create directly.
int rth_scene_desc(void* p, rt_scene_desc* out) { return guard([&] { *out = cam(p)->flat_scene().desc(); }); }

// Host-only helpers (no GPU needed): known answers of the construction substrate.
void rth_seed_next(uint64_t* state4, uint32_t n, uint64_t* out) {
    Seed s{{state4[0], state4[1], state4[2], state4[3]}};
    for (uint32_t i = 0; i < n; ++i) out[i] = s.next();
    std::memcpy(state4, s.s, sizeof(s.s));
}
void rth_seed_next_f32(uint64_t* state4, uint32_t n, float* out) {
    Seed s{{state4[0], state4[1], state4[2], state4[3]}};
    for (uint32_t i = 0; i < n; ++i) out[i] = s.next_f32();
    std::memcpy(state4, s.s, sizeof(s.s));
}
void rth_raytracing_camera(uint32_t w, uint32_t h, uint32_t depth, float vfov, const float* from, const float* at, const float* vup,
                           float defocus, float focus, const float* bg, rt_camera_desc* out) {
    *out = raytracing_camera(w, h, depth, vfov, v3(from), v3(at), v3(vup), defocus, focus, v3(bg));
}

}  // extern "C"
