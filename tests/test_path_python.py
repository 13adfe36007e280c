"""ray_colour and render (src/camera.rs:269-306,324-365) restated a second time in Python on top of the other independent
restatements (camera, traversal semantics, leaf hits, Material::scatter, the counter-based streams): for a small scene of spheres,
quads, a placed box, a light and a medium, the oracle's accumulated image must be this image (to 1e-4 on at least 95 % of the
pixels: a path that grazes something may take another turn where an ulp differs)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
import oracle_binding as ob  # noqa: E402
from test_camera_python import PyCamera  # noqa: E402
from test_hit_python import V, f32  # noqa: E402
from test_material_python import scatter  # noqa: E402
from test_oracle_kat import counter_stream  # noqa: E402
from test_traversal_python import hit  # noqa: E402


def tag(node, material):
    """The recursion of test_traversal_python.py returns no material: find it again by asking each tagged subtree alone."""
    return (node, material)


def closest(objects, o, d, time, key):
    best, best_mat, tmax = None, None, f32(np.inf)
    for node, material in objects:
        r = hit(node, o, d, time, f32(0.001), tmax, key)
        if r:
            best, best_mat, tmax = r, material, r[0]
    return best, best_mat


def ray_colour(cam, objects, background, max_depth, pixel, sample, seed):
    o, d, time = cam.ray(pixel, sample, seed)
    multiplier = V(1, 1, 1)
    for bounce in range(max_depth):
        key = (seed, pixel, sample, bounce)
        rec, material = closest(objects, o, d, time, key)
        if rec is None:
            return multiplier * background
        kind, param, colour = material
        emitted = colour if kind == "light" else V(0, 0, 0)
        draws = iter(f32(f32(r >> 8) * f32(1.0 / 16777216.0)) for r in counter_stream(seed, pixel, sample, bounce, 1, 96))
        out = scatter(kind, param, d, rec[2], rec[3], draws)
        if out is None:
            return multiplier * emitted
        multiplier = multiplier * colour + emitted
        o, d = rec[1], out
    return V(0, 0, 0)


def test_image_against_an_independent_restatement():
    w, h, depth, seed, spp = 10, 7, 5, 13, 2
    o = ob.OracleScene()
    specs = [("lambertian", 0.0, V(0.8, 0.3, 0.3)), ("metal", 0.2, V(0.7, 0.7, 0.9)), ("dielectric", 1.5, V(1, 1, 1)), ("light", 0.0, V(6, 5, 4)),
             ("lambertian", 0.0, V(0.3, 0.7, 0.3))]
    mats = []
    for kind, param, colour in specs:
        t = o.tex_solid(*(float(c) for c in colour))
        mats.append({"lambertian": lambda: o.lambertian(t), "metal": lambda: o.metal(t, float(param)), "dielectric": lambda: o.dielectric(t, float(param)),
                     "light": lambda: o.diffuse_light(t)}[kind]())
    fog_colour = V(0.9, 0.9, 0.9)
    fog_tex = o.tex_solid(0.9, 0.9, 0.9)
    objects, handles = [], []

    def add(handle, node, material):
        handles.append(handle)
        objects.append(tag(node, material))

    add(o.sphere((0, -100.5, 0), 100.0, mats[0]), ("sphere", V(0, -100.5, 0), f32(100.0)), specs[0])
    add(o.sphere((-1.2, 0, 0), 0.5, mats[1]), ("sphere", V(-1.2, 0, 0), f32(0.5)), specs[1])
    add(o.sphere((0, 0, 0.3), 0.5, mats[2]), ("sphere", V(0, 0, 0.3), f32(0.5)), specs[2])
    add(o.quad((-1, 1.2, -1), (2, 0, 0), (0, 0, 2), mats[3]), ("quad", V(-1, 1.2, -1), V(2, 0, 0), V(0, 0, 2)), specs[3])
    add(o.moving_sphere((1.3, 0, 0), (1.3, 0.3, 0), 0.4, mats[4]), ("moving", V(1.3, 0, 0), V(1.3, 0.3, 0), f32(0.4)), specs[4])
    rads = f32(f32(25.0) * f32(0.017453292519943295))
    box = o.move(o.rotate_y(o.box((0, 0, 0), (0.5, 0.7, 0.5), mats[4]), 25.0), (0.4, -0.5, -1.2))
    a, b = V(0, 0, 0), V(0.5, 0.7, 0.5)
    dx, dy, dz = V(b[0], 0, 0), V(0, b[1], 0), V(0, 0, b[2])
    quads = [(V(0, 0, b[2]), dx, dy), (V(b[0], 0, b[2]), -dz, dy), (V(b[0], 0, 0), -dx, dy), (V(0, 0, 0), dz, dy), (V(0, b[1], b[2]), dx, -dz), (V(0, 0, 0), dx, dz)]
    add(box, ("move", ("rotate_y", ("list", [("quad", *q) for q in quads]), np.sin(rads), np.cos(rads)), V(0.4, -0.5, -1.2)), specs[4])
    add(o.medium(o.sphere((-0.4, 0.1, -1.0), 0.45, mats[0]), 1.5, fog_tex), ("medium", ("sphere", V(-0.4, 0.1, -1.0), f32(0.45)), f32(f32(-1.0) / f32(1.5)), 0),
        ("isotropic", 0.0, fog_colour))
    o.set_world(o.bvh(handles))
    cam = dict(pos=(0.2, 0.6, -4.0), vfov=35.0, vup=(0, 1, 0), defocus_angle=0.4, focus_dist=4.0, pitch=-6.0, yaw=90.0)
    background = V(0.5, 0.6, 0.8)
    o.camera(cam["pos"], w, h, max_depth=depth, vfov=cam["vfov"], vup=cam["vup"], defocus_angle=cam["defocus_angle"], focus_dist=cam["focus_dist"],
             background=tuple(float(c) for c in background), pitch=cam["pitch"], yaw=cam["yaw"])
    mine = PyCamera(w, h, cam["vfov"], cam["pos"], cam["vup"], cam["defocus_angle"], cam["focus_dist"], cam["pitch"], cam["yaw"])
    want = np.zeros((h, w, 3), np.float32)
    with np.errstate(all="ignore"):
        for sample in range(1, spp + 1):
            for pixel in range(w * h):
                want[pixel // w, pixel % w] += ray_colour(mine, objects, background, depth, pixel, sample, seed)
    got = o.render(spp, seed=seed)[..., :3]
    close = np.all(np.abs(got - want) <= 1e-4 * np.maximum(1.0, np.abs(want)), axis=-1)
    assert close.mean() >= 0.95, (close.mean(), np.argwhere(~close)[:5])
    assert want.std() > 0.1
