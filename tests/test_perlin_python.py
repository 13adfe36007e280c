"""PerlinNoise::new, noise and turbulance (src/perlin_noise.rs:17-122) restated a second time, independently, in Python with numpy
float32 scalars, straight from the Rust text: the oracle's tables must be these tables and its noise these values.  (The host
library's tables are compared with the oracle's in test_host_scene.py, the kernels' noise with the oracle's on the GPU.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
import oracle_binding as ob  # noqa: E402

M = (1 << 64) - 1
f32 = np.float32


def rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M


class Seed:  # src/rng.rs:17-59
    def __init__(self, s):
        self.s = list(s)

    def next(self):
        s = self.s
        result = (rotl((s[0] + s[3]) & M, 23) + s[0]) & M
        t = (s[1] << 17) & M
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 45)
        return result

    def next_f32(self):
        return f32((1.0 / float(1 << 53)) * float(self.next() >> 11))

    def next_f32_range(self, lo, hi):
        return f32(f32(lo) + f32(f32(hi) - f32(lo)) * self.next_f32())


def dot(a, b):  # f32x4 product, ordered reduce_sum (lane 3 is 0)
    return f32(f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2])) + f32(0.0))


def random_unit(seed):  # src/math/vec3.rs:45-67
    while True:
        p = [seed.next_f32_range(-1.0, 1.0) for _ in range(3)]
        if dot(p, p) < f32(1.0):
            inv = f32(f32(1.0) / np.sqrt(dot(p, p)))
            return [f32(inv * c) for c in p]


def permutation(seed, n):  # src/perlin_noise.rs:106-122
    p = list(range(n))
    for i in range(n):
        target = seed.next() % (i + 1)
        p[i], p[target] = p[target], p[i]
    return p


def perlin_new(seed4, n):
    seed = Seed(seed4)
    vecs = [random_unit(seed) for _ in range(n)]
    return vecs, permutation(seed, n), permutation(seed, n), permutation(seed, n)


def noise(tables, p):  # src/perlin_noise.rs:39-87
    vecs, px, py, pz = tables
    n = len(vecs)
    fl = [f32(np.floor(c)) for c in p]
    u, v, w = (f32(c - f) for c, f in zip(p, fl))
    i, j, k = (int(f) for f in fl)
    c = [[[vecs[(px[((i + di) & 255) % n] ^ py[((j + dj) & 255) % n] ^ pz[((k + dk) & 255) % n]) % n] for dk in range(2)] for dj in range(2)] for di in range(2)]
    uu, vv, ww = (f32(f32(x * x) * f32(f32(3.0) - f32(f32(2.0) * x))) for x in (u, v, w))
    accum = f32(0.0)
    for a in range(2):
        for b in range(2):
            for d in range(2):
                weight_v = [f32(u - f32(a)), f32(v - f32(b)), f32(w - f32(d))]
                fa = f32(f32(f32(a) * uu) + f32(f32(f32(1.0) - f32(a)) * f32(f32(1.0) - uu)))
                fb = f32(f32(f32(b) * vv) + f32(f32(f32(1.0) - f32(b)) * f32(f32(1.0) - vv)))
                fd = f32(f32(f32(d) * ww) + f32(f32(f32(1.0) - f32(d)) * f32(f32(1.0) - ww)))
                accum = f32(accum + f32(f32(f32(fa * fb) * fd) * dot(c[a][b][d], weight_v)))
    return accum


def turbulance(tables, p, depth):  # src/perlin_noise.rs:90-102
    accum, weight, temp = f32(0.0), f32(1.0), [f32(c) for c in p]
    for _ in range(depth):
        accum = f32(accum + f32(weight * noise(tables, temp)))
        weight = f32(weight * f32(0.5))
        temp = [f32(c * f32(2.0)) for c in temp]
    return f32(abs(accum))


def test_perlin_tables_and_noise_against_an_independent_restatement():
    # scene 3 (simple_light, src/main.rs:503): PerlinNoise::new(arena, &mut Seed([1, 2, 3, 4]), 100)
    o = ob.OracleScene.builtin(3, 16, 16)
    assert o.perlin_count == 1
    vecs, px, py, pz = o.perlin_tables(0)
    mine = perlin_new([1, 2, 3, 4], 100)
    assert np.array_equal(np.array(mine[0], np.float32), vecs)
    assert mine[1] == px.tolist() and mine[2] == py.tolist() and mine[3] == pz.tolist()
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.normal(size=(150, 3)) * 3, rng.normal(size=(50, 3)) * 400, np.round(rng.normal(size=(30, 3)) * 5),
                          -np.abs(rng.normal(size=(30, 3))) * 1e4]).astype(np.float32)
    got_noise, got_turb = o.perlin_eval(0, pts, depth=7)
    with np.errstate(over="ignore"):
        want_noise = np.array([noise(mine, p) for p in pts], np.float32)
        want_turb = np.array([turbulance(mine, p, 7) for p in pts], np.float32)
    assert np.array_equal(got_noise, want_noise)
    assert np.array_equal(got_turb, want_turb)
