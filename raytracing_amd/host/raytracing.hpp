// raytracing.hpp — host-side scene-building API, mirroring the public API of the
// reference crate `raytracing-improved` (Camera, Hittable/BVH, Material, Texture,
// PerlinNoise, Seed, World) so that scene code reads like the reference's
// src/main.rs.  Nothing here traces rays: Camera::{empty,realtime,hdr}_render
// flatten the scene once (raytracing.cpp) and call the C ABI of include/rt_abi.h,
// which runs the render loop on the GPU.
#pragma once

#include <cstddef>
#include <cstdint>
#include <deque>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

#include "../../include/rt_abi.h"

namespace raytracing {

// ---- src/math/vec3.rs ------------------------------------------------------------
struct Vec3 {
    float x = 0, y = 0, z = 0;
    constexpr Vec3() = default;
    constexpr Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    static constexpr Vec3 ZERO() { return Vec3(0, 0, 0); }
    static constexpr Vec3 ONE() { return Vec3(1, 1, 1); }
};
using Point = Vec3;
using Colour = Vec3;
Vec3 operator+(Vec3 a, Vec3 b);
Vec3 operator-(Vec3 a, Vec3 b);
Vec3 operator-(Vec3 a);
Vec3 operator*(Vec3 a, Vec3 b);
Vec3 operator*(float s, Vec3 a);
Vec3 operator/(Vec3 a, float s);
bool operator==(Vec3 a, Vec3 b);
float dot(Vec3 a, Vec3 b);
Vec3 cross(Vec3 a, Vec3 b);
float length(Vec3 a);
Vec3 unit(Vec3 a);

// ---- src/math/interval.rs --------------------------------------------------------
struct Interval {
    float min, max;
    static constexpr Interval make(float lo, float hi) { return Interval{lo, hi}; }
};

// ---- src/rng.rs ------------------------------------------------------------------
// xoshiro256++; scene construction draws from it exactly as the reference does.
struct Seed {
    uint64_t s[4];
    uint64_t next();
    float next_f32();
    float next_f32_range(Interval r);
};
Vec3 random(Seed& seed);                       // Vec3::random
Vec3 random_range(Seed& seed, Interval r);     // Vec3::random_range
Vec3 random_unit(Seed& seed);                  // Vec3::random_unit

// ---- src/perlin_noise.rs ---------------------------------------------------------
struct PerlinTables {
    std::vector<float> rand_floats;            // 3 * point_count
    std::vector<uint32_t> perm_x, perm_y, perm_z;
};
class Arena;
struct PerlinNoise {
    const PerlinTables* tables = nullptr;
    static PerlinNoise make(Arena& arena, Seed& seed, size_t point_count);  // PerlinNoise::new
};

// image::Rgb32FImage stand-in: f32 or 8-bit RGB texels, row-major, top row first.
struct Rgb32FImage {
    uint32_t width = 0, height = 0;
    std::vector<float> rgb32f;    // either this ...
    std::vector<uint8_t> rgb8;    // ... or this is filled
};

// ---- src/texture.rs --------------------------------------------------------------
struct Texture {
    enum Kind { SolidColour, Checkerboard, Image, NoiseTexture, NotFound } kind = NotFound;
    Colour solid;
    float inv_scale = 0;
    const Texture* even = nullptr;
    const Texture* odd = nullptr;
    const Rgb32FImage* img = nullptr;
    PerlinNoise noise_tables;
    float scale = 0;
    static Texture colour(Colour c);
    static Texture checkerboard(float scale, const Texture* even, const Texture* odd);
    static Texture image(const Rgb32FImage* image);
    static Texture noise(PerlinNoise noise, float scale);
};

// ---- src/material.rs -------------------------------------------------------------
struct MaterialId {
    uint32_t v = 0;  // MaterialId::DEFAULT
};
struct Material {
    enum Kind { Lambertian, Metal, Dielectric, DiffuseLight, Isotropic, NotFound } kind = NotFound;
    Texture texture;
    float param = 0;  // fuzz_radius | refraction_index
    static Material lambertian(Texture t);
    static Material metal(Texture t, float fuzz_radius);
    static Material dielectric(Texture t, float refraction_index);
    static Material diffuse_light(Texture t);
    static Material isotropic(Texture t);
};
class MaterialMap {
  public:
    MaterialMap();  // id 0 = black lambertian (src/material.rs:40-44)
    MaterialId push(Material m);
    const Material& get(MaterialId id) const;
    size_t size() const { return map_.size(); }
  private:
    std::vector<Material> map_;
};

// ---- src/hittable.rs -------------------------------------------------------------
struct Sphere {
    Point centre; float radius; MaterialId material; uint32_t serial;
    static Sphere make(Point centre, float radius, MaterialId material);           // Sphere::new
};
struct MovingSphere {
    Point centre0; Vec3 dcentre; float radius; MaterialId material; uint32_t serial;
    static MovingSphere make(Point c1, Point c2, float radius, MaterialId material);  // MovingSphere::new
};
struct Quad {
    Point q; Vec3 u, v; MaterialId material; uint32_t serial;
    static Quad make(Point q, Vec3 u, Vec3 v, MaterialId material);                  // Quad::new
};
struct Hittable;
struct ConstantMedium {
    MaterialId phase_function; const Hittable* boundary; float neg_inv_density; uint32_t serial;
    static ConstantMedium make(MaterialMap& mmap, const Hittable* boundary, float density, Texture texture);  // ::new
};

struct Hittable {
    enum Kind { SphereKind, MovingSphereKind, QuadKind, ConstantMediumKind, Group, Move, RotateY } kind = Group;
    Sphere sphere_{};
    MovingSphere moving_{};
    Quad quad_{};
    ConstantMedium medium_{};
    std::vector<const Hittable*> children;  // Group: the list handed to Hittable::bvh
    const Hittable* obj = nullptr;          // Move / RotateY
    Vec3 offset;                            // Move
    float sin = 0, cos = 1;                 // RotateY

    static Hittable sphere(Sphere s);
    static Hittable quad(Quad q);
    static Hittable moving_sphere(MovingSphere s);
    static Hittable constant_medium(ConstantMedium m);
    static Hittable box_of_quads(Arena& arena, Point a, Point b, MaterialId mat);
    // Hittable::bvh: on this backend a plain group — the GPU library builds its own
    // flattened BVH over all leaves at set_world time.
    static Hittable bvh(Arena& arena, const std::vector<Hittable>& hittables);
    Hittable move_by(Arena& arena, Vec3 offset) const;
    Hittable rotate_y_by(Arena& arena, float degrees) const;
};

// sti::arena::Arena stand-in: owns everything scenes point to.
class Arena {
  public:
    const Hittable* alloc_new(Hittable h) { hittables_.push_back(std::move(h)); return &hittables_.back(); }
    const Texture* alloc_new(Texture t) { textures_.push_back(t); return &textures_.back(); }
    const Rgb32FImage* alloc_new(Rgb32FImage im) { images_.push_back(std::move(im)); return &images_.back(); }
    const PerlinTables* alloc_new(PerlinTables t) { perlins_.push_back(std::move(t)); return &perlins_.back(); }
  private:
    std::deque<Hittable> hittables_;
    std::deque<Texture> textures_;
    std::deque<Rgb32FImage> images_;
    std::deque<PerlinTables> perlins_;
};

// ---- src/lib.rs:17-26 ------------------------------------------------------------
struct World {
    const Hittable* entry = nullptr;
    MaterialMap material_map;
    static World make(const Hittable* entry, MaterialMap mmap) { return World{entry, std::move(mmap)}; }
};

// Flat copy of a World in the C ABI's terms; owns the arrays rt_scene_desc points to.
struct FlatScene {
    std::vector<rt_sphere> spheres;
    std::vector<rt_moving_sphere> moving_spheres;
    std::vector<rt_quad> quads;
    std::vector<rt_medium> media;
    std::vector<rt_prim_ref> world, boundary;
    std::vector<rt_transform> transforms;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_image> images;
    std::vector<rt_perlin> perlins;
    rt_scene_desc desc() const;
};
FlatScene flatten(const World& world);

class RenderError : public std::runtime_error {
  public:
    using std::runtime_error::runtime_error;
};

// RaytracingCamera::new (src/camera.rs:222-266): the viewport constants handed to the GPU.
rt_camera_desc raytracing_camera(size_t width, size_t height, size_t max_depth, float vfov, Vec3 look_from, Vec3 look_at, Vec3 vup,
                                 float defocus_angle, float focus_dist, Colour background_colour);

// The image file of `--image` (src/main.rs:171-183): RGBA f32, rows top to bottom.
void write_exr(const std::string& path, size_t width, size_t height, const float* rgba);
void write_pfm(const std::string& path, size_t width, size_t height, const float* rgba);

// ---- src/camera.rs:10-202 --------------------------------------------------------
class Camera {
  public:
    // Camera::new; `device` selects the GPU the render loop runs on.
    Camera(Arena& arena, Vec3 position, Vec3 direction, std::pair<size_t, size_t> resolution, float display_scale, size_t max_depth,
           float vfov, Vec3 vup, float defocus_angle, float focus_dist, Colour background_colour, int device = 0);
    ~Camera();
    Camera(const Camera&) = delete;
    Camera& operator=(const Camera&) = delete;

    void set_exposure(float exposure);
    void set_world(World world);
    void empty_render();                          // one more sample per pixel
    // One more sample, tone-mapped 0BGR words.  Like the reference's `&[u32]`, the words live in the camera (in page-locked
    // memory, so that the download is one DMA transfer) and are valid until its next render call.
    struct Words {
        const uint32_t* ptr = nullptr;
        size_t len = 0;
        const uint32_t* data() const { return ptr; }
        size_t size() const { return len; }
        const uint32_t& operator[](size_t i) const { return ptr[i]; }
        const uint32_t* begin() const { return ptr; }
        const uint32_t* end() const { return ptr + len; }
    };
    Words realtime_render();
    std::vector<Vec3> hdr_render();               // one more sample, sqrt(mean)
    // Not in the reference: `n` consecutive empty_render calls in one GPU submission.
    void render_samples(uint32_t n);
    // `--image` (src/main.rs:157-186): sample_count - 1 empty_render calls and one hdr_render; returns the
    // W*H RGBA floats (alpha 1) and, unless `path` is empty, writes them as OpenEXR (or PFM for "*.pfm").
    std::vector<float> render_to_image(size_t sample_count, const std::string& path);
    // Not in the reference: the accumulation buffer, sample count and seed to a file and back, so that a
    // long render can be continued by another process (the sample numbers carry on where they stopped).
    void save_checkpoint(const std::string& path);
    void load_checkpoint(const std::string& path);
    void change_pitch_yaw_by(float delta_pitch, float delta_yaw);
    void move_by(Vec3 step);
    Vec3 forward() const { return direction_; }
    Vec3 backward() const { return -forward(); }
    Vec3 left() const { return -right(); }
    Vec3 right() const { return unit(cross(forward(), up())); }
    Vec3 up() const { return Vec3(0.0f, 1.0f, 0.0f); }
    float exposure() const { return exposure_; }
    size_t samples() const { return samples_; }
    float pitch() const { return pitch_; }
    float yaw() const { return yaw_; }
    float display_scale() const { return display_scale_; }
    std::pair<size_t, size_t> render_resolution() const { return render_resolution_; }
    std::pair<size_t, size_t> display_resolution() const;

    // Backend access (not in the reference)
    void set_seed(uint64_t seed) { seed_ = seed; }
    const rt_camera_desc& raytracing_camera_desc();   // refreshed like update_raytracing_camera
    rt_ctx* context() { ensure_backend(); return ctx_; }
    rt_scene* scene() { ensure_backend(); return scene_; }
    const FlatScene& flat_scene() const { return flat_; }
    float* device_accumulator() { ensure_backend(); return d_acc_; }
    std::vector<float> download_accumulator();        // W*H*4 floats
    void upload_accumulator(const float* acc, size_t samples);  // replaces the buffer: `samples` samples per pixel are in it

  private:
    void ensure_backend();
    void update_raytracing_camera();
    void force_update_raytracing_camera();
    void check(int status, const char* what);

    Vec3 position_, direction_;
    float display_scale_;
    std::pair<size_t, size_t> render_resolution_;
    float pitch_ = 0, yaw_ = 0, vfov_;
    Vec3 vup_;
    float focus_dist_, defocus_angle_;
    size_t max_depth_;
    rt_camera_desc rt_cam_{};
    uint32_t* final_colours_ = nullptr;  // page-locked, one word per pixel (created with the backend)
    size_t samples_ = 0;
    World world_;
    Colour background_colour_;
    float exposure_ = 1.0f;
    uint64_t seed_ = 0;

    int device_ = 0;
    bool scene_dirty_ = true;
    rt_ctx* ctx_ = nullptr;
    rt_scene* scene_ = nullptr;
    FlatScene flat_;
    float* d_acc_ = nullptr;     // acc_colours, resident on the GPU
    void* d_out_ = nullptr;      // output of the finalize kernel
};

// The reference's scene builders (src/main.rs:291-731).  id 0..8 = `--scene`;
// 100 earth + perlin spheres, 101 Cornell box with a glass sphere, 102 N random spheres.
struct SceneOptions {
    size_t width = 0, height = 0;   // 0 = the reference's size at resolution scale 100
    size_t max_depth = 25;
    float display_scale = 1.0f;
    const Rgb32FImage* earth = nullptr;  // decoded earthmap.jpg for scenes 4, 8, 100
    uint64_t stress_count = 100000;
    int device = 0;
};
std::unique_ptr<Camera> build_scene(Arena& arena, int id, const SceneOptions& opts);

}  // namespace raytracing
