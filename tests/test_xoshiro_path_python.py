"""The reference's own random numbers: one xoshiro256++ stream per image row, seeded [y, n_samples, W * H, y] and consumed in
program order - jitter x, jitter y, [lens disk trials], time, then per bounce the draws of Material::scatter
(src/camera.rs:282-304,311-320,339-364).  Restated in Python on top of the other independent restatements, for a scene without
media (a medium's draw comes wherever the BVH's visiting order puts it): the oracle's reference-faithful mode must accumulate the
same image."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
import oracle_binding as ob  # noqa: E402
from test_camera_python import PyCamera, dot  # noqa: E402
from test_hit_python import V, f32  # noqa: E402
from test_material_python import scatter  # noqa: E402
from test_path_python import closest  # noqa: E402
from test_perlin_python import Seed  # noqa: E402


def draws_of(seed):
    while True:
        yield seed.next_f32()


def get_ray(cam, draws, x, y):
    centre = cam.p00 + f32(x) * cam.du + f32(y) * cam.dv
    px, py = f32(f32(-0.5) + next(draws)), f32(f32(-0.5) + next(draws))
    sample_pt = centre + (px * cam.du + py * cam.dv)
    origin = cam.centre
    if cam.defocus_angle > 0:
        while True:     # random_in_unit_disk: next_f32_range(-1, 1) twice per trial
            a, b = f32(f32(-1.0) + f32(f32(2.0) * next(draws))), f32(f32(-1.0) + f32(f32(2.0) * next(draws)))
            if dot(V(a, b, 0), V(a, b, 0)) < f32(1.0):
                break
        origin = cam.centre + a * cam.ddu + b * cam.ddv
    return origin, sample_pt - origin, next(draws)


def test_reference_random_numbers_against_an_independent_restatement():
    w, h, depth, spp = 9, 6, 5, 3
    o = ob.OracleScene()
    specs = [("lambertian", 0.0, V(0.8, 0.3, 0.3)), ("metal", 0.2, V(0.7, 0.7, 0.9)), ("dielectric", 1.5, V(1, 1, 1)), ("light", 0.0, V(6, 5, 4)),
             ("isotropic", 0.0, V(0.3, 0.7, 0.3))]
    mats = []
    for kind, param, colour in specs:
        t = o.tex_solid(*(float(c) for c in colour))
        mats.append({"lambertian": lambda: o.lambertian(t), "metal": lambda: o.metal(t, float(param)), "dielectric": lambda: o.dielectric(t, float(param)),
                     "light": lambda: o.diffuse_light(t), "isotropic": lambda: o.isotropic(t)}[kind]())
    objects, handles = [], []
    for handle, node, material in (
            (o.sphere((0, -100.5, 0), 100.0, mats[0]), ("sphere", V(0, -100.5, 0), f32(100.0)), specs[0]),
            (o.sphere((-1.2, 0, 0), 0.5, mats[1]), ("sphere", V(-1.2, 0, 0), f32(0.5)), specs[1]),
            (o.sphere((0, 0, 0.3), 0.5, mats[2]), ("sphere", V(0, 0, 0.3), f32(0.5)), specs[2]),
            (o.quad((-1, 1.2, -1), (2, 0, 0), (0, 0, 2), mats[3]), ("quad", V(-1, 1.2, -1), V(2, 0, 0), V(0, 0, 2)), specs[3]),
            (o.moving_sphere((1.3, 0, 0), (1.3, 0.3, 0), 0.4, mats[4]), ("moving", V(1.3, 0, 0), V(1.3, 0.3, 0), f32(0.4)), specs[4])):
        handles.append(handle)
        objects.append((node, material))
    o.set_world(o.bvh(handles))
    cam = dict(pos=(0.2, 0.6, -4.0), vfov=35.0, vup=(0, 1, 0), defocus_angle=0.4, focus_dist=4.0, pitch=-6.0, yaw=90.0)
    background = V(0.5, 0.6, 0.8)
    o.camera(cam["pos"], w, h, max_depth=depth, vfov=cam["vfov"], vup=cam["vup"], defocus_angle=cam["defocus_angle"], focus_dist=cam["focus_dist"],
             background=tuple(float(c) for c in background), pitch=cam["pitch"], yaw=cam["yaw"])
    mine = PyCamera(w, h, cam["vfov"], cam["pos"], cam["vup"], cam["defocus_angle"], cam["focus_dist"], cam["pitch"], cam["yaw"])
    want = np.zeros((h, w, 3), np.float32)
    with np.errstate(all="ignore"):
        for n_samples in range(1, spp + 1):          # render(world, n_samples, ..): the sample's number is part of the seed
            for y in range(h):
                draws = draws_of(Seed([y, n_samples, w * h, y]))
                for x in range(w):
                    origin, direction, time = get_ray(mine, draws, x, y)
                    multiplier, colour_out = V(1, 1, 1), V(0, 0, 0)
                    for bounce in range(depth):
                        rec, material = closest(objects, origin, direction, time, None)
                        if rec is None:
                            colour_out = multiplier * background
                            break
                        kind, param, colour = material
                        emitted = colour if kind == "light" else V(0, 0, 0)
                        out = scatter(kind, param, direction, rec[2], rec[3], draws)
                        if out is None:
                            colour_out = multiplier * emitted
                            break
                        multiplier = multiplier * colour + emitted
                        origin, direction = rec[1], out
                    want[y, x] += colour_out
    got = o.render(spp, mode=ob.RNG_XOSHIRO, threads=1)[..., :3]
    close = np.all(np.abs(got - want) <= 1e-4 * np.maximum(1.0, np.abs(want)), axis=-1)
    assert close.mean() >= 0.9, (close.mean(), np.argwhere(~close)[:5])
