// raytracing.cpp — host-side mirror of the reference crate's construction API
// (see raytracing.hpp): numeric substrate, constructors, the tree flattener and
// the Camera that drives the GPU through the C ABI.  Compiled with
// -ffp-contract=off so that scene construction (sphere centres, perlin tables,
// camera basis) rounds exactly as the reference's f32 code does.
#include "raytracing.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <string>

namespace raytracing {

// ---- src/math/vec3.rs ------------------------------------------------------------
Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
Vec3 operator/(Vec3 a, float s) { return (1.0f / s) * a; }  // :222-228
bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
float dot(Vec3 a, Vec3 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + 0.0f; }  // ordered reduce_sum, :105-107
Vec3 cross(Vec3 l, Vec3 r) { return {l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x}; }
float length(Vec3 a) { return std::sqrt(dot(a, a)); }
Vec3 unit(Vec3 a) { return a / length(a); }

static float to_radians(float deg) { return deg * 0.017453292519943295769236907684886f; }

// ---- src/rng.rs:23-59 --------------------------------------------------------------
static uint64_t rotl(uint64_t x, uint64_t k) { return (x << k) | (x >> (64 - k)); }
uint64_t Seed::next() {
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
float Seed::next_f32() {
    const double scale = 1.0 / (double)(1ull << 53);
    return (float)(scale * (double)(next() >> 11));
}
float Seed::next_f32_range(Interval r) { return r.min + (r.max - r.min) * next_f32(); }
Vec3 random(Seed& seed) { Vec3 v; v.x = seed.next_f32(); v.y = seed.next_f32(); v.z = seed.next_f32(); return v; }
Vec3 random_range(Seed& seed, Interval r) {
    Vec3 v; v.x = seed.next_f32_range(r); v.y = seed.next_f32_range(r); v.z = seed.next_f32_range(r); return v;
}
Vec3 random_unit(Seed& seed) {  // src/math/vec3.rs:57-67
    for (;;) {
        Vec3 p = random_range(seed, Interval{-1.0f, 1.0f});
        if (dot(p, p) < 1.0f) return unit(p);
    }
}

// ---- src/perlin_noise.rs:17-37,106-122 ---------------------------------------------
static void generate_perm(Seed& seed, std::vector<uint32_t>& p, size_t n) {
    p.resize(n);
    for (size_t i = 0; i < n; ++i) p[i] = (uint32_t)i;
    for (size_t i = 0; i < n; ++i) {
        size_t target = (size_t)(seed.next() % (uint64_t)(i + 1));
        std::swap(p[i], p[target]);
    }
}
PerlinNoise PerlinNoise::make(Arena& arena, Seed& seed, size_t point_count) {
    PerlinTables t;
    for (size_t i = 0; i < point_count; ++i) {
        Vec3 v = random_unit(seed);
        t.rand_floats.push_back(v.x); t.rand_floats.push_back(v.y); t.rand_floats.push_back(v.z);
    }
    generate_perm(seed, t.perm_x, point_count);
    generate_perm(seed, t.perm_y, point_count);
    generate_perm(seed, t.perm_z, point_count);
    return PerlinNoise{arena.alloc_new(std::move(t))};
}

// ---- src/texture.rs:76-96 ----------------------------------------------------------
Texture Texture::colour(Colour c) { Texture t; t.kind = SolidColour; t.solid = c; return t; }
Texture Texture::checkerboard(float scale, const Texture* even, const Texture* odd) {
    Texture t; t.kind = Checkerboard; t.inv_scale = 1.0f / scale; t.even = even; t.odd = odd; return t;
}
Texture Texture::image(const Rgb32FImage* image) { Texture t; t.kind = Image; t.img = image; return t; }
Texture Texture::noise(PerlinNoise noise, float scale) { Texture t; t.kind = NoiseTexture; t.noise_tables = noise; t.scale = scale; return t; }

// ---- src/material.rs:34-49,125-151 -------------------------------------------------
Material Material::lambertian(Texture t) { Material m; m.kind = Lambertian; m.texture = t; return m; }
Material Material::metal(Texture t, float fuzz_radius) { Material m; m.kind = Metal; m.texture = t; m.param = fuzz_radius; return m; }
Material Material::dielectric(Texture t, float ri) { Material m; m.kind = Dielectric; m.texture = t; m.param = ri; return m; }
Material Material::diffuse_light(Texture t) { Material m; m.kind = DiffuseLight; m.texture = t; return m; }
Material Material::isotropic(Texture t) { Material m; m.kind = Isotropic; m.texture = t; return m; }
MaterialMap::MaterialMap() { map_.push_back(Material::lambertian(Texture::colour(Colour::ZERO()))); }
MaterialId MaterialMap::push(Material m) { map_.push_back(m); return MaterialId{(uint32_t)map_.size() - 1}; }
const Material& MaterialMap::get(MaterialId id) const { return map_.at(id.v); }

// ---- src/hittable.rs ---------------------------------------------------------------
static std::atomic<uint32_t> g_serial{0};  // creation order of leaves: source of rt_*.prim_id
Sphere Sphere::make(Point c, float r, MaterialId m) { return Sphere{c, r, m, g_serial.fetch_add(1)}; }
MovingSphere MovingSphere::make(Point c1, Point c2, float r, MaterialId m) { return MovingSphere{c1, c2 - c1, r, m, g_serial.fetch_add(1)}; }
Quad Quad::make(Point q, Vec3 u, Vec3 v, MaterialId m) { return Quad{q, u, v, m, g_serial.fetch_add(1)}; }
ConstantMedium ConstantMedium::make(MaterialMap& mmap, const Hittable* boundary, float density, Texture texture) {
    MaterialId phase = mmap.push(Material::isotropic(texture));  // :420-424
    return ConstantMedium{phase, boundary, -1.0f / density, g_serial.fetch_add(1)};
}
Hittable Hittable::sphere(Sphere s) { Hittable h; h.kind = SphereKind; h.sphere_ = s; return h; }
Hittable Hittable::quad(Quad q) { Hittable h; h.kind = QuadKind; h.quad_ = q; return h; }
Hittable Hittable::moving_sphere(MovingSphere s) { Hittable h; h.kind = MovingSphereKind; h.moving_ = s; return h; }
Hittable Hittable::constant_medium(ConstantMedium m) { Hittable h; h.kind = ConstantMediumKind; h.medium_ = m; return h; }
Hittable Hittable::box_of_quads(Arena& arena, Point a, Point b, MaterialId mat) {  // :81-100
    Point mn(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z));
    Point mx(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z));
    Vec3 dx(mx.x - mn.x, 0, 0), dy(0, mx.y - mn.y, 0), dz(0, 0, mx.z - mn.z);
    std::vector<Hittable> sides;
    sides.push_back(Hittable::quad(Quad::make(Point(mn.x, mn.y, mx.z), dx, dy, mat)));   // front
    sides.push_back(Hittable::quad(Quad::make(Point(mx.x, mn.y, mx.z), -dz, dy, mat)));  // right
    sides.push_back(Hittable::quad(Quad::make(Point(mx.x, mn.y, mn.z), -dx, dy, mat)));  // back
    sides.push_back(Hittable::quad(Quad::make(Point(mn.x, mn.y, mn.z), dz, dy, mat)));   // left
    sides.push_back(Hittable::quad(Quad::make(Point(mn.x, mx.y, mx.z), dx, -dz, mat)));  // top
    sides.push_back(Hittable::quad(Quad::make(Point(mn.x, mn.y, mn.z), dx, dz, mat)));   // bottom
    return Hittable::bvh(arena, sides);
}
Hittable Hittable::bvh(Arena& arena, const std::vector<Hittable>& hittables) {
    Hittable h;
    h.kind = Group;
    for (const Hittable& x : hittables) h.children.push_back(arena.alloc_new(x));
    return h;
}
Hittable Hittable::move_by(Arena& arena, Vec3 off) const {  // :227-231
    Hittable h; h.kind = Move; h.obj = arena.alloc_new(*this); h.offset = off; return h;
}
Hittable Hittable::rotate_y_by(Arena& arena, float degrees) const {  // :234-243
    float rads = to_radians(degrees);
    Hittable h; h.kind = RotateY; h.obj = arena.alloc_new(*this); h.sin = std::sin(rads); h.cos = std::cos(rads); return h;
}

// ---- flattening: Hittable tree + MaterialMap -> rt_scene_desc ---------------------------
namespace {

struct Placement {
    bool any = false;
    float s = 0, c = 1;
    Vec3 off;
};
// object-space point of the wrapped object = R * (p - off); see rt_transform in rt_abi.h
Placement compose_move(const Placement& outer, Vec3 offset) {
    Placement r = outer;
    r.any = true;
    Vec3 inv(outer.c * offset.x + outer.s * offset.z, offset.y, -outer.s * offset.x + outer.c * offset.z);
    r.off = outer.off + inv;
    return r;
}
Placement compose_rot(const Placement& outer, float s, float c) {
    Placement r = outer;
    r.any = true;
    r.s = s * outer.c + c * outer.s;
    r.c = c * outer.c - s * outer.s;
    return r;
}

struct Flattener {
    FlatScene& out;
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> leaf_index[3];  // (serial, transform) -> index, per leaf kind
    std::map<const Rgb32FImage*, uint32_t> image_index;
    std::map<const PerlinTables*, uint32_t> perlin_index;

    explicit Flattener(FlatScene& o) : out(o) {}

    uint32_t transform_of(const Placement& p) {
        if (!p.any) return RT_NO_TRANSFORM;
        for (uint32_t i = 0; i < out.transforms.size(); ++i) {
            const rt_transform& t = out.transforms[i];
            if (t.sin_theta == p.s && t.cos_theta == p.c && t.offset[0] == p.off.x && t.offset[1] == p.off.y && t.offset[2] == p.off.z) return i;
        }
        rt_transform t{};
        t.sin_theta = p.s; t.cos_theta = p.c; t.offset[0] = p.off.x; t.offset[1] = p.off.y; t.offset[2] = p.off.z;
        out.transforms.push_back(t);
        return (uint32_t)out.transforms.size() - 1;
    }

    // refs: where leaf references go (world list or a medium's boundary list)
    void walk(const Hittable* h, const Placement& pl, std::vector<rt_prim_ref>& refs, bool in_boundary) {
        switch (h->kind) {
            case Hittable::SphereKind: {
                uint32_t xf = transform_of(pl);
                auto key = std::make_pair(h->sphere_.serial, xf);
                auto it = leaf_index[0].find(key);
                if (it == leaf_index[0].end()) {
                    rt_sphere s{};
                    s.centre[0] = h->sphere_.centre.x; s.centre[1] = h->sphere_.centre.y; s.centre[2] = h->sphere_.centre.z;
                    s.radius = h->sphere_.radius; s.material = h->sphere_.material.v; s.prim_id = h->sphere_.serial; s.transform = xf;
                    out.spheres.push_back(s);
                    it = leaf_index[0].emplace(key, (uint32_t)out.spheres.size() - 1).first;
                }
                refs.push_back(rt_prim_ref{RT_PRIM_SPHERE, it->second});
                break;
            }
            case Hittable::MovingSphereKind: {
                uint32_t xf = transform_of(pl);
                auto key = std::make_pair(h->moving_.serial, xf);
                auto it = leaf_index[1].find(key);
                if (it == leaf_index[1].end()) {
                    rt_moving_sphere s{};
                    s.centre0[0] = h->moving_.centre0.x; s.centre0[1] = h->moving_.centre0.y; s.centre0[2] = h->moving_.centre0.z;
                    s.dcentre[0] = h->moving_.dcentre.x; s.dcentre[1] = h->moving_.dcentre.y; s.dcentre[2] = h->moving_.dcentre.z;
                    s.radius = h->moving_.radius; s.material = h->moving_.material.v; s.prim_id = h->moving_.serial; s.transform = xf;
                    out.moving_spheres.push_back(s);
                    it = leaf_index[1].emplace(key, (uint32_t)out.moving_spheres.size() - 1).first;
                }
                refs.push_back(rt_prim_ref{RT_PRIM_MOVING_SPHERE, it->second});
                break;
            }
            case Hittable::QuadKind: {
                uint32_t xf = transform_of(pl);
                auto key = std::make_pair(h->quad_.serial, xf);
                auto it = leaf_index[2].find(key);
                if (it == leaf_index[2].end()) {
                    rt_quad q{};
                    q.q[0] = h->quad_.q.x; q.q[1] = h->quad_.q.y; q.q[2] = h->quad_.q.z;
                    q.u[0] = h->quad_.u.x; q.u[1] = h->quad_.u.y; q.u[2] = h->quad_.u.z;
                    q.v[0] = h->quad_.v.x; q.v[1] = h->quad_.v.y; q.v[2] = h->quad_.v.z;
                    q.material = h->quad_.material.v; q.prim_id = h->quad_.serial; q.transform = xf;
                    out.quads.push_back(q);
                    it = leaf_index[2].emplace(key, (uint32_t)out.quads.size() - 1).first;
                }
                refs.push_back(rt_prim_ref{RT_PRIM_QUAD, it->second});
                break;
            }
            case Hittable::ConstantMediumKind: {
                if (in_boundary) throw RenderError("a ConstantMedium cannot be the boundary of another ConstantMedium on this backend");
                std::vector<rt_prim_ref> b;
                walk(h->medium_.boundary, pl, b, true);
                rt_medium m{};
                m.first_boundary = (uint32_t)out.boundary.size();
                m.n_boundary = (uint32_t)b.size();
                m.neg_inv_density = h->medium_.neg_inv_density;
                m.phase_material = h->medium_.phase_function.v;
                m.prim_id = h->medium_.serial;
                out.boundary.insert(out.boundary.end(), b.begin(), b.end());
                out.media.push_back(m);
                refs.push_back(rt_prim_ref{RT_PRIM_MEDIUM, (uint32_t)out.media.size() - 1});
                break;
            }
            case Hittable::Group:
                for (const Hittable* c : h->children) walk(c, pl, refs, in_boundary);
                break;
            case Hittable::Move: walk(h->obj, compose_move(pl, h->offset), refs, in_boundary); break;
            case Hittable::RotateY: walk(h->obj, compose_rot(pl, h->sin, h->cos), refs, in_boundary); break;
        }
    }

    uint32_t add_texture(const Texture& t) {
        rt_texture r{};
        switch (t.kind) {
            case Texture::SolidColour:
                r.kind = RT_TEX_SOLID; r.colour[0] = t.solid.x; r.colour[1] = t.solid.y; r.colour[2] = t.solid.z; break;
            case Texture::Checkerboard:
                r.kind = RT_TEX_CHECKER; r.scale = t.inv_scale; r.even = add_texture(*t.even); r.odd = add_texture(*t.odd); break;
            case Texture::Image: {
                r.kind = RT_TEX_IMAGE;
                auto it = image_index.find(t.img);
                if (it == image_index.end()) {
                    rt_image im{};
                    im.width = t.img->width; im.height = t.img->height;
                    if (!t.img->rgb32f.empty()) { im.format = RT_IMAGE_RGB32F; im.texels = t.img->rgb32f.data(); }
                    else { im.format = RT_IMAGE_RGB8; im.texels = t.img->rgb8.data(); }
                    out.images.push_back(im);
                    it = image_index.emplace(t.img, (uint32_t)out.images.size() - 1).first;
                }
                r.image = it->second;
                break;
            }
            case Texture::NoiseTexture: {
                r.kind = RT_TEX_NOISE; r.scale = t.scale;
                const PerlinTables* pt = t.noise_tables.tables;
                auto it = perlin_index.find(pt);
                if (it == perlin_index.end()) {
                    rt_perlin p{};
                    p.point_count = (uint32_t)pt->perm_x.size();
                    p.rand_vecs = pt->rand_floats.data(); p.perm_x = pt->perm_x.data(); p.perm_y = pt->perm_y.data(); p.perm_z = pt->perm_z.data();
                    out.perlins.push_back(p);
                    it = perlin_index.emplace(pt, (uint32_t)out.perlins.size() - 1).first;
                }
                r.perlin = it->second;
                break;
            }
            case Texture::NotFound:  // Texture::value returns black for NotFound (src/texture.rs:71)
                r.kind = RT_TEX_SOLID; break;
        }
        out.textures.push_back(r);
        return (uint32_t)out.textures.size() - 1;
    }
};

}  // namespace

rt_scene_desc FlatScene::desc() const {
    rt_scene_desc d{};
    d.abi_version = RT_ABI_VERSION;
    d.n_spheres = (uint32_t)spheres.size(); d.spheres = spheres.data();
    d.n_moving_spheres = (uint32_t)moving_spheres.size(); d.moving_spheres = moving_spheres.data();
    d.n_quads = (uint32_t)quads.size(); d.quads = quads.data();
    d.n_media = (uint32_t)media.size(); d.media = media.data();
    d.n_world = (uint32_t)world.size(); d.world = world.data();
    d.n_boundary = (uint32_t)boundary.size(); d.boundary = boundary.data();
    d.n_transforms = (uint32_t)transforms.size(); d.transforms = transforms.data();
    d.n_materials = (uint32_t)materials.size(); d.materials = materials.data();
    d.n_textures = (uint32_t)textures.size(); d.textures = textures.data();
    d.n_images = (uint32_t)images.size(); d.images = images.data();
    d.n_perlins = (uint32_t)perlins.size(); d.perlins = perlins.data();
    return d;
}

FlatScene flatten(const World& world) {
    FlatScene out;
    Flattener f(out);
    for (size_t i = 0; i < world.material_map.size(); ++i) {
        const Material& m = world.material_map.get(MaterialId{(uint32_t)i});
        if (m.kind == Material::NotFound) throw RenderError("MaterialKind::NotFound cannot be rendered (unimplemented!() in the reference)");
        rt_material r{};
        r.kind = (uint32_t)m.kind;
        r.param = m.param;
        r.texture = f.add_texture(m.texture);
        out.materials.push_back(r);
    }
    if (world.entry) f.walk(world.entry, Placement{}, out.world, false);
    // prim_id = rank of the leaf's creation serial within this world
    std::vector<uint32_t> serials;
    for (auto& s : out.spheres) serials.push_back(s.prim_id);
    for (auto& s : out.moving_spheres) serials.push_back(s.prim_id);
    for (auto& q : out.quads) serials.push_back(q.prim_id);
    for (auto& m : out.media) serials.push_back(m.prim_id);
    std::sort(serials.begin(), serials.end());
    serials.erase(std::unique(serials.begin(), serials.end()), serials.end());
    auto rank = [&](uint32_t s) { return (uint32_t)(std::lower_bound(serials.begin(), serials.end(), s) - serials.begin()); };
    for (auto& s : out.spheres) s.prim_id = rank(s.prim_id);
    for (auto& s : out.moving_spheres) s.prim_id = rank(s.prim_id);
    for (auto& q : out.quads) q.prim_id = rank(q.prim_id);
    for (auto& m : out.media) m.prim_id = rank(m.prim_id);
    return out;
}

// ---- src/camera.rs:222-266 -----------------------------------------------------------
rt_camera_desc raytracing_camera(size_t width, size_t height, size_t max_depth, float vfov, Vec3 look_from, Vec3 look_at, Vec3 vup,
                                 float defocus_angle, float focus_dist, Colour background_colour) {
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
    Vec3 ddu = defocus_radius * u, ddv = defocus_radius * v;
    rt_camera_desc c{};
    c.width = (uint32_t)width; c.height = (uint32_t)height; c.max_depth = (uint32_t)max_depth; c.defocus_angle = defocus_angle;
    auto st = [](float* d, Vec3 s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    st(c.centre, centre); st(c.pixel00_loc, pixel00_loc); st(c.pixel_delta_u, pixel_delta_u); st(c.pixel_delta_v, pixel_delta_v);
    st(c.defocus_disk_u, ddu); st(c.defocus_disk_v, ddv); st(c.background_colour, background_colour);
    return c;
}

// ---- src/camera.rs:36-202 ------------------------------------------------------------
void Camera::check(int status, const char* what) {
    if (status != RT_OK) throw RenderError(std::string(what) + ": " + rt_last_error());
}

Camera::Camera(Arena& arena, Vec3 position, Vec3 direction, std::pair<size_t, size_t> resolution, float display_scale, size_t max_depth,
               float vfov, Vec3 vup, float defocus_angle, float focus_dist, Colour background_colour, int device)
    : position_(position), direction_(direction), display_scale_(display_scale), render_resolution_(resolution), vfov_(vfov), vup_(vup),
      focus_dist_(focus_dist), defocus_angle_(defocus_angle), max_depth_(max_depth), background_colour_(background_colour) {
    rt_cam_ = raytracing_camera(resolution.first, resolution.second, max_depth, vfov, position, position + direction, vup, defocus_angle,
                                focus_dist, background_colour);
    device_ = device;
    // the reference starts with a unit sphere of the default material (:53-56)
    set_world(World::make(arena.alloc_new(Hittable::sphere(Sphere::make(Point::ZERO(), 1.0f, MaterialId{}))), MaterialMap()));
}

// The GPU context, the device copy of the scene and the device buffers are
// created on first use, so that scenes can be built and inspected without a GPU.
void Camera::ensure_backend() {
    size_t n = render_resolution_.first * render_resolution_.second;
    if (!ctx_) {
        check(rt_ctx_create(device_, &ctx_), "rt_ctx_create");
        check(rt_buffer_create(ctx_, n * 4 * sizeof(float), (void**)&d_acc_), "rt_buffer_create");
        check(rt_buffer_create(ctx_, n * 4 * sizeof(float), &d_out_), "rt_buffer_create");
        check(rt_buffer_zero(ctx_, d_acc_, n * 4 * sizeof(float)), "rt_buffer_zero");
        check(rt_host_buffer_create(ctx_, n * sizeof(uint32_t), (void**)&final_colours_), "rt_host_buffer_create");
        std::memset(final_colours_, 0, n * sizeof(uint32_t));
    }
    if (scene_dirty_) {
        rt_scene_desc d = flat_.desc();
        rt_scene* fresh = nullptr;
        check(rt_scene_create(ctx_, &d, &fresh), "rt_scene_create");
        if (scene_) rt_scene_destroy(scene_);
        scene_ = fresh;
        scene_dirty_ = false;
    }
}

Camera::~Camera() {
    if (scene_) rt_scene_destroy(scene_);
    if (ctx_) {
        rt_buffer_destroy(ctx_, d_acc_);
        rt_buffer_destroy(ctx_, d_out_);
        rt_host_buffer_destroy(ctx_, final_colours_);
        rt_ctx_destroy(ctx_);
    }
}

void Camera::set_exposure(float exposure) {  // :66-69
    force_update_raytracing_camera();
    exposure_ = exposure;
}

void Camera::set_world(World world) {  // :74-77
    samples_ = 0;
    world_ = std::move(world);
    flat_ = flatten(world_);
    scene_dirty_ = true;
}

void Camera::force_update_raytracing_camera() {  // :133-145
    Vec3 direction(std::cos(to_radians(yaw_)) * std::cos(to_radians(pitch_)), std::sin(to_radians(pitch_)),
                   std::sin(to_radians(yaw_)) * std::cos(to_radians(pitch_)));
    rt_cam_ = raytracing_camera(render_resolution_.first, render_resolution_.second, max_depth_, vfov_, position_, position_ + direction,
                                vup_, defocus_angle_, focus_dist_, background_colour_);
}

void Camera::update_raytracing_camera() {  // :125-130
    ensure_backend();
    if (samples_ != 0) return;
    force_update_raytracing_camera();
    check(rt_buffer_zero(ctx_, d_acc_, render_resolution_.first * render_resolution_.second * 4 * sizeof(float)), "rt_buffer_zero");
}

const rt_camera_desc& Camera::raytracing_camera_desc() {
    if (samples_ == 0) force_update_raytracing_camera();
    return rt_cam_;
}

void Camera::render_samples(uint32_t n) {
    if (n == 0) return;
    update_raytracing_camera();
    uint32_t first = (uint32_t)samples_ + 1;  // the reference passes the 1-based sample number (:82-83)
    check(rt_render_accumulate_device(scene_, &rt_cam_, seed_, first, n, d_acc_), "rt_render_accumulate_device");
    samples_ += n;  // only once the samples are on their way: a refused call leaves the count as it was
}

void Camera::empty_render() { render_samples(1); }  // :80-85

Camera::Words Camera::realtime_render() {  // :89-104
    render_samples(1);
    size_t n = render_resolution_.first * render_resolution_.second;
    check(rt_finalize_device(ctx_, d_acc_, n, (uint32_t)samples_, RT_FINALIZE_TONEMAP_RGBA, exposure_, d_out_), "rt_finalize_device");
    check(rt_buffer_download(ctx_, final_colours_, d_out_, n * sizeof(uint32_t)), "rt_buffer_download");
    return Words{final_colours_, n};
}

std::vector<Vec3> Camera::hdr_render() {  // :107-122
    render_samples(1);
    size_t n = render_resolution_.first * render_resolution_.second;
    check(rt_finalize_device(ctx_, d_acc_, n, (uint32_t)samples_, RT_FINALIZE_SQRT, exposure_, d_out_), "rt_finalize_device");
    std::vector<float> tmp(n * 4);
    check(rt_buffer_download(ctx_, tmp.data(), d_out_, n * 4 * sizeof(float)), "rt_buffer_download");
    std::vector<Vec3> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = Vec3(tmp[4 * i], tmp[4 * i + 1], tmp[4 * i + 2]);
    return out;
}

std::vector<float> Camera::download_accumulator() {
    ensure_backend();
    size_t n = render_resolution_.first * render_resolution_.second;
    std::vector<float> acc(n * 4);
    check(rt_buffer_download(ctx_, acc.data(), d_acc_, n * 4 * sizeof(float)), "rt_buffer_download");
    return acc;
}

void Camera::upload_accumulator(const float* acc, size_t samples) {
    ensure_backend();
    force_update_raytracing_camera();
    size_t n = render_resolution_.first * render_resolution_.second;
    check(rt_buffer_upload(ctx_, d_acc_, acc, n * 4 * sizeof(float)), "rt_buffer_upload");
    samples_ = samples;
}

void Camera::change_pitch_yaw_by(float delta_pitch, float delta_yaw) {  // :149-160
    if (delta_pitch == 0.0f && delta_yaw == 0.0f) return;
    pitch_ += delta_pitch;
    yaw_ += delta_yaw;
    direction_ = Vec3(std::cos(to_radians(yaw_)) * std::cos(to_radians(pitch_)), std::sin(to_radians(pitch_)),
                      std::sin(to_radians(yaw_)) * std::cos(to_radians(pitch_)));
    samples_ = 0;
}

void Camera::move_by(Vec3 step) {  // :163-168
    position_ = position_ + step;
    if (!(step == Vec3::ZERO())) samples_ = 0;
}

std::pair<size_t, size_t> Camera::display_resolution() const {  // :197-201
    return {(size_t)((float)render_resolution_.first * display_scale_), (size_t)((float)render_resolution_.second * display_scale_)};
}

}  // namespace raytracing
