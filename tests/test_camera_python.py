"""RaytracingCamera::new and get_ray (src/camera.rs:133-146,222-266,311-320,368-382) restated a second time in Python (float64
arithmetic rounded to f32 after every operation, numpy's trigonometry): the oracle's primary rays must be these rays to a few
ulps - numpy's cos / sin / tan are not glibc's - for cameras of every orientation, with and without defocus blur."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
import oracle_binding as ob  # noqa: E402
from test_oracle_kat import counter_stream  # noqa: E402

f32 = np.float32


def V(x, y, z):
    return np.array([x, y, z], np.float32)


def dot(a, b):
    return f32(f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2])) + f32(0.0))


def cross(l, r):
    return V(f32(l[1] * r[2]) - f32(l[2] * r[1]), f32(l[2] * r[0]) - f32(l[0] * r[2]), f32(l[0] * r[1]) - f32(l[1] * r[0]))


def unit(a):
    return (f32(1.0) / np.sqrt(dot(a, a))) * a


def div(a, s):  # Vec3 / f32 is (1 / s) * v (src/math/vec3.rs:222-228)
    return (f32(1.0) / f32(s)) * a


def radians(x):
    return f32(f32(x) * f32(0.017453292519943295))


class PyCamera:
    def __init__(self, width, height, vfov, pos, vup, defocus_angle, focus_dist, pitch, yaw):
        pos, vup = V(*pos), V(*vup)
        cp, sp = np.cos(radians(pitch)), np.sin(radians(pitch))
        direction = V(f32(np.cos(radians(yaw)) * cp), sp, f32(np.sin(radians(yaw)) * cp))
        look_at = pos + direction
        focus_dist = f32(focus_dist)
        h = np.tan(f32(radians(vfov) / f32(2.0)))
        viewport_height = f32(f32(f32(2.0) * h) * focus_dist)
        viewport_width = f32(viewport_height * f32(f32(width) / f32(height)))
        w = unit(pos - look_at)
        u = unit(cross(vup, w))
        v = cross(w, u)
        viewport_u, viewport_v = viewport_width * u, viewport_height * -v
        self.du, self.dv = div(viewport_u, width), div(viewport_v, height)
        upper_left = pos - focus_dist * w - div(viewport_u, 2.0) - div(viewport_v, 2.0)
        self.p00 = upper_left + f32(0.5) * (self.du + self.dv)
        radius = f32(focus_dist * np.tan(radians(f32(f32(defocus_angle) / f32(2.0)))))
        self.ddu, self.ddv = radius * u, radius * v
        self.centre, self.defocus_angle, self.width = pos, f32(defocus_angle), width

    def ray(self, pixel, sample, seed):
        draws = iter(f32(f32(r >> 8) * f32(1.0 / 16777216.0)) for r in counter_stream(seed, pixel, sample, 0, 0, 64))
        x, y = pixel % self.width, pixel // self.width
        centre = self.p00 + f32(x) * self.du + f32(y) * self.dv
        px, py = f32(f32(-0.5) + next(draws)), f32(f32(-0.5) + next(draws))
        sample_pt = centre + (px * self.du + py * self.dv)
        origin = self.centre
        if self.defocus_angle > 0:
            while True:
                a, b = f32(f32(-1.0) + f32(f32(2.0) * next(draws))), f32(f32(-1.0) + f32(f32(2.0) * next(draws)))
                if dot(V(a, b, 0), V(a, b, 0)) < f32(1.0):
                    break
            origin = self.centre + a * self.ddu + b * self.ddv
        return origin, sample_pt - origin, next(draws)


def test_primary_rays_against_an_independent_restatement():
    rng = np.random.default_rng(1)
    for k in range(12):
        c = dict(pos=tuple(float(x) for x in rng.uniform(-50, 50, 3)), width=int(rng.integers(2, 40)), height=int(rng.integers(2, 30)),
                 vfov=float(rng.choice([1.0, 20.0, 40.0, 90.0, 150.0])), vup=tuple(float(x) for x in rng.normal(size=3)),
                 defocus_angle=float(rng.choice([0.0, 0.6, 10.0])), focus_dist=float(rng.choice([0.1, 10.0, 1000.0])),
                 pitch=float(rng.choice([0.0, -60.0, 15.0, 45.0])), yaw=float(rng.uniform(-360, 360)))
        o = ob.OracleScene()
        o.set_world(o.bvh([o.sphere((0, 0, 0), 1.0, o.lambertian(o.tex_solid(0.5, 0.5, 0.5)))]))
        o.camera(c["pos"], c["width"], c["height"], max_depth=3, vfov=c["vfov"], vup=c["vup"], defocus_angle=c["defocus_angle"],
                 focus_dist=c["focus_dist"], background=(1, 1, 1), pitch=c["pitch"], yaw=c["yaw"])
        mine = PyCamera(c["width"], c["height"], c["vfov"], c["pos"], c["vup"], c["defocus_angle"], c["focus_dist"], c["pitch"], c["yaw"])
        pix = rng.integers(0, c["width"] * c["height"], 12).astype(np.uint32)
        for sample in (1, 9):
            rays = o.primary_rays(pix, sample, seed=5)
            for r, p in zip(rays, pix):
                origin, direction, time = mine.ray(int(p), sample, 5)
                size = max(1.0, float(np.abs(direction).max()), float(np.abs(origin).max()))
                assert np.abs(r["origin"] - origin).max() <= 2e-5 * size, (c, p)
                assert np.abs(r["direction"] - direction).max() <= 2e-5 * size, (c, p)
                assert r["time"] == time
