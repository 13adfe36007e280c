// scenes.cpp — the reference's nine scenes (src/main.rs:291-731) written against
// the host API of raytracing.hpp, plus the benchmark configurations that have no
// single builder in the reference (ids 100..102).  The xoshiro draw order is the
// reference's, so geometry, materials and perlin tables come out bit-identical.
#include <cmath>

#include "raytracing.hpp"

namespace raytracing {
namespace {

std::pair<size_t, size_t> resolution(const SceneOptions& o, size_t dw, size_t dh) {
    return {o.width ? o.width : dw, o.height ? o.height : dh};
}

Texture checker(Arena& arena, float scale) {
    return Texture::checkerboard(scale, arena.alloc_new(Texture::colour(Colour::ZERO())), arena.alloc_new(Texture::colour(Colour::ONE())));
}

// src/main.rs:553-612 (night = false) and :615-684 (night = true)
std::unique_ptr<Camera> bouncing_spheres(Arena& arena, const SceneOptions& opts, bool night) {
    std::vector<Hittable> world;
    MaterialMap mmap;
    MaterialId material_ground = mmap.push(Material::lambertian(checker(arena, 0.64f)));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, -1000.0f, 0.0f), 1000.0f, material_ground)));
    Seed seed{{69, 420, 420, 69}};
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            float choose_mat = seed.next_f32();
            float cx = (float)a + 9.0f * seed.next_f32();
            float cz = (float)b + 9.0f * seed.next_f32();
            Vec3 centre(cx, 0.2f, cz);
            Vec3 centre_2 = centre + Vec3(0.0f, seed.next_f32() * 0.2f, 0.0f);
            if (length(centre - Point(4.0f, 0.2f, 0.0f)) <= 0.9f) continue;
            Material mat;
            if (choose_mat < (night ? 0.5f : 0.8f)) {
                Colour c1 = random(seed);
                Colour c2 = random(seed);
                mat = Material::lambertian(Texture::colour(c1 * c2));
            } else if (night && choose_mat < 0.8f) {
                Colour c1 = random(seed);
                Colour c2 = random(seed);
                mat = Material::diffuse_light(Texture::colour(25.0f * (c1 * c2)));
            } else if (choose_mat < 0.95f) {
                Colour albedo = random_range(seed, Interval{0.5f, 1.0f});
                float fuzz = seed.next_f32_range(Interval{0.0f, 0.5f});
                mat = Material::metal(Texture::colour(albedo), fuzz);
            } else {
                mat = Material::dielectric(Texture::colour(Colour::ONE()), 1.5f);
            }
            MaterialId id = mmap.push(mat);
            Hittable hittable = Hittable::moving_sphere(MovingSphere::make(centre, centre_2, 0.2f, id));
            if (night && seed.next_f32() < 0.125f) {
                PerlinNoise noise = PerlinNoise::make(arena, seed, 64);
                Texture texture = Texture::noise(noise, seed.next_f32() * 2.0f);
                const Hittable* boundary = arena.alloc_new(hittable);
                float density = seed.next_f32();
                hittable = Hittable::constant_medium(ConstantMedium::make(mmap, boundary, density, texture));
            }
            world.push_back(hittable);
        }
    }
    MaterialId m1 = mmap.push(Material::dielectric(Texture::colour(Colour(1.0f, 1.0f, 1.0f)), 1.5f));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, 1.0f, 0.0f), 1.0f, m1)));
    MaterialId m2 = mmap.push(Material::lambertian(Texture::colour(Colour(0.4f, 0.2f, 0.1f))));
    world.push_back(Hittable::sphere(Sphere::make(Point(-4.0f, 1.0f, 0.0f), 1.0f, m2)));
    MaterialId m3 = mmap.push(Material::metal(Texture::colour(Colour(0.7f, 0.6f, 0.5f)), 0.0f));
    Hittable last = Hittable::sphere(Sphere::make(Point(4.0f, 1.0f, 0.0f), 1.0f, m3));
    world.push_back(night ? last.move_by(arena, Vec3::ONE()) : last);

    Hittable root = Hittable::bvh(arena, world);
    auto camera = std::make_unique<Camera>(arena, Point(-10.0f, 5.0f, -10.0f), Vec3(1.0f, 0.0f, 0.0f), resolution(opts, 1600, 900),
                                           opts.display_scale, opts.max_depth, 20.0f, Vec3(0.0f, 2.0f, 0.0f), 0.0f, 10.0f,
                                           night ? Colour(0.0f, 0.0f, 0.0f) : Colour(0.7f, 0.7f, 0.7f), opts.device);
    camera->set_world(World::make(arena.alloc_new(root), std::move(mmap)));
    camera->change_pitch_yaw_by(-15.0f, 45.0f);
    return camera;
}

// src/main.rs:523-550
std::unique_ptr<Camera> quads(Arena& arena, const SceneOptions& opts) {
    std::vector<Hittable> world;
    MaterialMap mmap;
    MaterialId left_red = mmap.push(Material::lambertian(Texture::colour(Colour(1.0f, 0.2f, 0.2f))));
    MaterialId back_green = mmap.push(Material::diffuse_light(Texture::colour(10.0f * Colour(0.2f, 1.0f, 0.2f))));
    MaterialId right_blue = mmap.push(Material::lambertian(Texture::colour(Colour(0.2f, 0.2f, 1.0f))));
    MaterialId upper_orange = mmap.push(Material::lambertian(Texture::colour(Colour(1.0f, 0.5f, 0.0f))));
    MaterialId lower_teal = mmap.push(Material::lambertian(Texture::colour(Colour(0.2f, 0.8f, 0.8f))));
    world.push_back(Hittable::quad(Quad::make(Point(-3.0f, -2.0f, 5.0f), Point(0.0f, 0.0f, -4.0f), Point(0.0f, 4.0f, 0.0f), left_red)));
    world.push_back(Hittable::quad(Quad::make(Point(-2.0f, -2.0f, 0.0f), Point(4.0f, 0.0f, 0.0f), Point(0.0f, 4.0f, 0.0f), back_green)));
    world.push_back(Hittable::quad(Quad::make(Point(3.0f, -2.0f, 1.0f), Point(0.0f, 0.0f, 4.0f), Point(0.0f, 4.0f, 0.0f), right_blue)));
    world.push_back(Hittable::quad(Quad::make(Point(-2.0f, 3.0f, 1.0f), Point(4.0f, 0.0f, 0.0f), Point(0.0f, 0.0f, 4.0f), upper_orange)));
    world.push_back(Hittable::quad(Quad::make(Point(-2.0f, -3.0f, 5.0f), Point(4.0f, 0.0f, 0.0f), Point(0.0f, 0.0f, -4.0f), lower_teal)));
    auto camera = std::make_unique<Camera>(arena, Point(0.0f, 0.0f, 9.0f), Vec3(1.0f, 0.0f, 0.0f), resolution(opts, 1000, 1000),
                                           opts.display_scale, opts.max_depth, 80.0f, Vec3(0.0f, 2.0f, 0.0f), 0.0f, 10.0f,
                                           Colour(0.0f, 0.0f, 0.0f), opts.device);
    camera->change_pitch_yaw_by(0.0f, -90.0f);
    camera->set_world(World::make(arena.alloc_new(Hittable::bvh(arena, world)), std::move(mmap)));
    return camera;
}

// src/main.rs:499-520
std::unique_ptr<Camera> simple_light(Arena& arena, const SceneOptions& opts) {
    std::vector<Hittable> world;
    MaterialMap mmap;
    Seed pseed{{1, 2, 3, 4}};
    Texture pertext = Texture::noise(PerlinNoise::make(arena, pseed, 100), 4.0f);
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, -1000.0f, 0.0f), 1000.0f, mmap.push(Material::lambertian(pertext)))));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, 2.0f, 0.0f), 2.0f, mmap.push(Material::lambertian(pertext)))));
    MaterialId diff_light = mmap.push(Material::diffuse_light(Texture::colour(Colour(4.0f, 4.0f, 4.0f))));
    world.push_back(Hittable::quad(Quad::make(Point(3.0f, 1.0f, -2.0f), Vec3(2.0f, 0.0f, 0.0f), Vec3(0.0f, 2.0f, 0.0f), diff_light)));
    auto camera = std::make_unique<Camera>(arena, Point(26.0f, 2.0f, 6.0f), Vec3(1.0f, 0.0f, 0.0f), resolution(opts, 1600, 900),
                                           opts.display_scale, opts.max_depth, 20.0f, Vec3(0.0f, 2.0f, 0.0f), 0.0f, 10.0f,
                                           Colour::ZERO(), opts.device);
    camera->change_pitch_yaw_by(0.0f, 200.0f);
    camera->set_world(World::make(arena.alloc_new(Hittable::bvh(arena, world)), std::move(mmap)));
    return camera;
}

// src/main.rs:687-712; the reference's "earthmap3.png" is not shipped, its earthmap.jpg stands in
std::unique_ptr<Camera> world_sphere(Arena& arena, const SceneOptions& opts) {
    std::vector<Hittable> world;
    MaterialMap mmap;
    MaterialId earth = mmap.push(Material::diffuse_light(Texture::image(opts.earth)));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, 0.0f, 0.0f), 2.0f, earth)));
    MaterialId ground = mmap.push(Material::lambertian(checker(arena, 1.0f)));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, -1002.0f, 0.0f), 1000.0f, ground)));
    Hittable root = Hittable::bvh(arena, world);
    auto camera = std::make_unique<Camera>(arena, Point(-10.0f, 5.0f, -10.0f), Vec3(1.0f, 0.0f, 0.0f), resolution(opts, 1600, 1000),
                                           opts.display_scale, opts.max_depth, 20.0f, Vec3(0.0f, 2.0f, 0.0f), 0.0f, 10.0f,
                                           Colour(0.98431373f, 0.56470588f, 0.38431373f), opts.device);
    camera->change_pitch_yaw_by(-20.0f, 40.0f);
    camera->set_world(World::make(arena.alloc_new(root), std::move(mmap)));
    return camera;
}

// src/main.rs:715-731
std::unique_ptr<Camera> checkered_spheres(Arena& arena, const SceneOptions& opts) {
    std::vector<Hittable> world;
    MaterialMap mmap;
    MaterialId ground = mmap.push(Material::lambertian(checker(arena, 1.0f)));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, -10.0f, 0.0f), 10.0f, ground)));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, 10.0f, 0.0f), 10.0f, ground)));
    Hittable root = Hittable::bvh(arena, world);
    auto camera = std::make_unique<Camera>(arena, Point(-25.0f, 0.0f, 0.0f), Vec3(1.0f, 0.0f, 0.0f), resolution(opts, 1600, 1000),
                                           opts.display_scale, opts.max_depth, 20.0f, Vec3(0.0f, 2.0f, 0.0f), 0.0f, 10.0f,
                                           Colour(0.7f, 0.7f, 0.7f), opts.device);
    camera->set_world(World::make(arena.alloc_new(root), std::move(mmap)));
    return camera;
}

struct CornellMaterials {
    MaterialId red, white, green, light;
};
CornellMaterials cornell_walls(std::vector<Hittable>& world, MaterialMap& mmap) {
    CornellMaterials m;
    m.red = mmap.push(Material::lambertian(Texture::colour(Colour(0.65f, 0.05f, 0.05f))));
    m.white = mmap.push(Material::lambertian(Texture::colour(Colour(0.73f, 0.73f, 0.73f))));
    m.green = mmap.push(Material::lambertian(Texture::colour(Colour(0.12f, 0.45f, 0.15f))));
    m.light = mmap.push(Material::diffuse_light(Texture::colour(Colour(15.0f, 15.0f, 15.0f))));
    world.push_back(Hittable::quad(Quad::make(Point(555.0f, 0.0f, 0.0f), Vec3(0.0f, 555.0f, 0.0f), Vec3(0.0f, 0.0f, 555.0f), m.green)));
    world.push_back(Hittable::quad(Quad::make(Point(0.0f, 0.0f, 0.0f), Vec3(0.0f, 555.0f, 0.0f), Vec3(0.0f, 0.0f, 555.0f), m.red)));
    world.push_back(Hittable::quad(Quad::make(Point(343.0f, 554.0f, 332.0f), Vec3(-130.0f, 0.0f, 0.0f), Vec3(0.0f, 0.0f, -105.0f), m.light)));
    world.push_back(Hittable::quad(Quad::make(Point(0.0f, 0.0f, 0.0f), Vec3(555.0f, 0.0f, 0.0f), Vec3(0.0f, 0.0f, 555.0f), m.white)));
    world.push_back(Hittable::quad(Quad::make(Point(555.0f, 555.0f, 555.0f), Vec3(-555.0f, 0.0f, 0.0f), Vec3(0.0f, 0.0f, -555.0f), m.white)));
    world.push_back(Hittable::quad(Quad::make(Point(0.0f, 0.0f, 555.0f), Vec3(555.0f, 0.0f, 0.0f), Vec3(0.0f, 555.0f, 0.0f), m.white)));
    return m;
}
std::unique_ptr<Camera> cornell_camera(Arena& arena, const SceneOptions& opts) {
    auto camera = std::make_unique<Camera>(arena, Point(278.0f, 278.0f, -800.0f), Vec3(1.0f, 0.0f, 0.0f), resolution(opts, 1000, 1000),
                                           opts.display_scale, opts.max_depth, 40.0f, Vec3(0.0f, 2.0f, 0.0f), 0.0f, 10.0f,
                                           Colour::ZERO(), opts.device);
    camera->change_pitch_yaw_by(0.0f, 90.0f);
    return camera;
}

// src/main.rs:413-448; glass_sphere swaps box2 for a dielectric sphere (id 101)
std::unique_ptr<Camera> cornell_box(Arena& arena, const SceneOptions& opts, bool glass_sphere) {
    std::vector<Hittable> world;
    MaterialMap mmap;
    CornellMaterials m = cornell_walls(world, mmap);
    Hittable box1 = Hittable::box_of_quads(arena, Point(0.0f, 0.0f, 0.0f), Point(165.0f, 330.0f, 165.0f), m.white)
                        .rotate_y_by(arena, 15.0f)
                        .move_by(arena, Vec3(265.0f, 0.0f, 295.0f));
    world.push_back(box1);
    if (!glass_sphere) {
        Hittable box2 = Hittable::box_of_quads(arena, Point(0.0f, 0.0f, 0.0f), Point(165.0f, 165.0f, 165.0f), m.white)
                            .rotate_y_by(arena, -18.0f)
                            .move_by(arena, Vec3(130.0f, 0.0f, 65.0f));
        world.push_back(box2);
    } else {
        MaterialId glass = mmap.push(Material::dielectric(Texture::colour(Colour::ONE()), 1.5f));
        world.push_back(Hittable::sphere(Sphere::make(Point(190.0f, 90.0f, 190.0f), 90.0f, glass)));
    }
    auto camera = cornell_camera(arena, opts);
    camera->set_world(World::make(arena.alloc_new(Hittable::bvh(arena, world)), std::move(mmap)));
    return camera;
}

// src/main.rs:451-496
std::unique_ptr<Camera> cornell_box_fog(Arena& arena, const SceneOptions& opts) {
    std::vector<Hittable> world;
    MaterialMap mmap;
    CornellMaterials m = cornell_walls(world, mmap);
    Hittable box1 = Hittable::box_of_quads(arena, Point(0.0f, 0.0f, 0.0f), Point(165.0f, 330.0f, 165.0f), m.white)
                        .rotate_y_by(arena, 15.0f)
                        .move_by(arena, Vec3(265.0f, 0.0f, 295.0f));
    Hittable fog1 = Hittable::constant_medium(ConstantMedium::make(mmap, arena.alloc_new(box1), 0.01f, Texture::colour(Colour::ZERO())));
    Hittable box2 = Hittable::box_of_quads(arena, Point(0.0f, 0.0f, 0.0f), Point(165.0f, 165.0f, 165.0f), m.white)
                        .rotate_y_by(arena, -18.0f)
                        .move_by(arena, Vec3(130.0f, 0.0f, 65.0f));
    Hittable fog2 = Hittable::constant_medium(ConstantMedium::make(mmap, arena.alloc_new(box2), 0.01f, Texture::colour(Colour::ONE())));
    world.push_back(fog1);
    world.push_back(fog2);
    auto camera = cornell_camera(arena, opts);
    camera->set_world(World::make(arena.alloc_new(Hittable::bvh(arena, world)), std::move(mmap)));
    return camera;
}

// src/main.rs:291-410
std::unique_ptr<Camera> the_final_scene(Arena& arena, const SceneOptions& opts) {
    Seed seed{{69, 420, 420, 69}};
    std::vector<Hittable> world;
    MaterialMap mmap;

    std::vector<Hittable> boxes_ground;
    MaterialId ground = mmap.push(Material::lambertian(Texture::colour(Colour(0.48f, 0.83f, 0.53f))));
    const int boxes_per_side = 20;
    for (int ii = 0; ii < boxes_per_side; ++ii) {
        float i = (float)ii;
        for (int jj = 0; jj < boxes_per_side; ++jj) {
            float j = (float)jj;
            float w = 100.0f;
            float x0 = -1000.0f + i * w;
            float z0 = -1000.0f + j * w;
            float y0 = 0.0f;
            float x1 = x0 + w;
            float y1 = seed.next_f32_range(Interval{1.0f, 101.0f});
            float z1 = z0 + w;
            boxes_ground.push_back(Hittable::box_of_quads(arena, Point(x0, y0, z0), Point(x1, y1, z1), ground));
        }
    }
    world.push_back(Hittable::bvh(arena, boxes_ground));

    MaterialId light = mmap.push(Material::diffuse_light(Texture::colour(Colour(7.0f, 7.0f, 7.0f))));
    world.push_back(Hittable::quad(Quad::make(Point(123.0f, 553.0f, 147.0f), Vec3(300.0f, 0.0f, 0.0f), Vec3(0.0f, 0.0f, 265.0f), light)));

    Point centre1(400.0f, 400.0f, 200.0f);
    Point centre2 = centre1 + Point(30.0f, 0.0f, 0.0f);
    MaterialId sphere_material = mmap.push(Material::lambertian(Texture::colour(Colour(0.7f, 0.3f, 0.1f))));
    world.push_back(Hittable::moving_sphere(MovingSphere::make(centre1, centre2, 50.0f, sphere_material)));

    MaterialId glass = mmap.push(Material::dielectric(Texture::colour(Colour::ONE()), 1.5f));
    world.push_back(Hittable::sphere(Sphere::make(Point(260.0f, 150.0f, 45.0f), 50.0f, glass)));
    MaterialId metal = mmap.push(Material::metal(Texture::colour(Colour(0.8f, 0.8f, 0.9f)), 1.0f));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, 150.0f, 145.0f), 50.0f, metal)));

    MaterialId shell1 = mmap.push(Material::dielectric(Texture::colour(Colour::ONE()), 1.5f));
    Hittable boundary = Hittable::sphere(Sphere::make(Point(360.0f, 150.0f, 145.0f), 70.0f, shell1));
    world.push_back(boundary);
    world.push_back(Hittable::constant_medium(ConstantMedium::make(mmap, arena.alloc_new(boundary), 0.2f, Texture::colour(Colour(0.2f, 0.4f, 0.9f)))));

    MaterialId shell2 = mmap.push(Material::dielectric(Texture::colour(Colour::ONE()), 1.5f));
    Hittable boundary2 = Hittable::sphere(Sphere::make(Point(0.0f, 0.0f, 0.0f), 5000.0f, shell2));
    world.push_back(boundary2);
    world.push_back(Hittable::constant_medium(ConstantMedium::make(mmap, arena.alloc_new(boundary2), 0.00005f, Texture::colour(Colour::ONE()))));

    MaterialId earth_material = mmap.push(Material::lambertian(Texture::image(opts.earth)));
    world.push_back(Hittable::sphere(Sphere::make(Point(400.0f, 200.0f, 400.0f), 100.0f, earth_material)));

    PerlinNoise perlin_noise = PerlinNoise::make(arena, seed, 256);
    MaterialId pertext = mmap.push(Material::lambertian(Texture::noise(perlin_noise, 0.2f)));
    world.push_back(Hittable::sphere(Sphere::make(Point(220.0f, 280.0f, 300.0f), 80.0f, pertext)));

    std::vector<Hittable> stress_balls;
    MaterialId white = mmap.push(Material::lambertian(Texture::colour(Colour::ONE())));
    Interval interval{0.0f, 165.0f};
    const int ns = 1000;
    for (int k = 0; k < ns; ++k) stress_balls.push_back(Hittable::sphere(Sphere::make(random_range(seed, interval), 10.0f, white)));
    world.push_back(Hittable::bvh(arena, stress_balls).rotate_y_by(arena, 15.0f).move_by(arena, Vec3(-100.0f, 270.0f, 395.0f)));

    auto camera = std::make_unique<Camera>(arena, Point(478.0f, 278.0f, -600.0f), Vec3(1.0f, 0.0f, 0.0f), resolution(opts, 1000, 1000),
                                           opts.display_scale, opts.max_depth, 40.0f, Vec3(0.0f, 1.0f, 0.0f), 0.0f, 10.0f,
                                           Colour::ZERO(), opts.device);
    camera->set_world(World::make(arena.alloc_new(Hittable::bvh(arena, world)), std::move(mmap)));
    camera->change_pitch_yaw_by(0.0f, 108.0f);
    return camera;
}

// id 100: scene 3's perlin setup (src/main.rs:503-505) and scene 8's earth material
// (src/main.rs:369-370) under a ceiling light.
std::unique_ptr<Camera> earth_and_perlin(Arena& arena, const SceneOptions& opts) {
    std::vector<Hittable> world;
    MaterialMap mmap;
    Seed pseed{{1, 2, 3, 4}};
    Texture pertext = Texture::noise(PerlinNoise::make(arena, pseed, 100), 4.0f);
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, -1000.0f, 0.0f), 1000.0f, mmap.push(Material::lambertian(pertext)))));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, 2.0f, 0.0f), 2.0f, mmap.push(Material::lambertian(pertext)))));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, 2.0f, 4.5f), 2.0f, mmap.push(Material::lambertian(pertext)))));
    MaterialId earth = mmap.push(Material::lambertian(Texture::image(opts.earth)));
    world.push_back(Hittable::sphere(Sphere::make(Point(0.0f, 2.0f, -4.5f), 2.0f, earth)));
    MaterialId diff_light = mmap.push(Material::diffuse_light(Texture::colour(Colour(4.0f, 4.0f, 4.0f))));
    world.push_back(Hittable::quad(Quad::make(Point(-4.0f, 7.0f, -6.0f), Vec3(8.0f, 0.0f, 0.0f), Vec3(0.0f, 0.0f, 12.0f), diff_light)));
    auto camera = std::make_unique<Camera>(arena, Point(26.0f, 3.0f, 6.0f), Vec3(1.0f, 0.0f, 0.0f), resolution(opts, 1920, 1080),
                                           opts.display_scale, opts.max_depth, 20.0f, Vec3(0.0f, 2.0f, 0.0f), 0.0f, 10.0f,
                                           Colour::ZERO(), opts.device);
    camera->change_pitch_yaw_by(-2.0f, 193.0f);
    camera->set_world(World::make(arena.alloc_new(Hittable::bvh(arena, world)), std::move(mmap)));
    return camera;
}

// id 102: N lambertian spheres, centres U[0,1000)^3, radii U[0.05,0.5), Seed([N,1,2,3])
std::unique_ptr<Camera> bvh_stress(Arena& arena, const SceneOptions& opts) {
    std::vector<Hittable> world;
    MaterialMap mmap;
    MaterialId grey = mmap.push(Material::lambertian(Texture::colour(Colour(0.5f, 0.5f, 0.5f))));
    Seed seed{{opts.stress_count, 1, 2, 3}};
    world.reserve(opts.stress_count);
    for (uint64_t i = 0; i < opts.stress_count; ++i) {
        Point c = random_range(seed, Interval{0.0f, 1000.0f});
        float r = seed.next_f32_range(Interval{0.05f, 0.5f});
        world.push_back(Hittable::sphere(Sphere::make(c, r, grey)));
    }
    auto camera = std::make_unique<Camera>(arena, Point(-800.0f, 500.0f, -800.0f), Vec3(1.0f, 0.0f, 0.0f), resolution(opts, 1920, 1080),
                                           opts.display_scale, opts.max_depth, 40.0f, Vec3(0.0f, 1.0f, 0.0f), 0.0f, 10.0f,
                                           Colour::ONE(), opts.device);
    camera->set_world(World::make(arena.alloc_new(Hittable::bvh(arena, world)), std::move(mmap)));
    camera->change_pitch_yaw_by(0.0f, 45.0f);
    return camera;
}

}  // namespace

std::unique_ptr<Camera> build_scene(Arena& arena, int id, const SceneOptions& opts) {
    bool needs_earth = id == 4 || id == 8 || id == 100;
    if (needs_earth && !opts.earth) throw RenderError("this scene needs the decoded earth map (SceneOptions::earth)");
    switch (id) {
        case 0: return bouncing_spheres(arena, opts, false);
        case 1: return bouncing_spheres(arena, opts, true);
        case 2: return quads(arena, opts);
        case 3: return simple_light(arena, opts);
        case 4: return world_sphere(arena, opts);
        case 5: return checkered_spheres(arena, opts);
        case 6: return cornell_box(arena, opts, false);
        case 7: return cornell_box_fog(arena, opts);
        case 8: return the_final_scene(arena, opts);
        case 100: return earth_and_perlin(arena, opts);
        case 101: return cornell_box(arena, opts, true);
        case 102: return bvh_stress(arena, opts);
        default: throw RenderError("scene number must be within 0..=8 (or 100..102)");
    }
}

}  // namespace raytracing
