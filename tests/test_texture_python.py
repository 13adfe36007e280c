"""Texture::value (src/texture.rs:40-73) restated a second time in Python: checker (of checkers), image lookup and the noise
texture on top of test_perlin_python.py's noise.  The oracle's values must be these values: bit for bit for checker and image,
to 1e-6 for the noise texture (numpy's sin is not glibc's)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
import oracle_binding as ob  # noqa: E402
from test_perlin_python import perlin_new, turbulance  # noqa: E402

f32 = np.float32


def rust_as_i32(x):
    if np.isnan(x):
        return 0
    return int(np.clip(float(x), -2147483648.0, 2147483647.0))


def rust_as_u32(x):
    if np.isnan(x):
        return 0
    return int(np.clip(float(x), 0.0, 4294967295.0))


def clamp01(x):  # f32::clamp hands NaN through
    return x if np.isnan(x) else min(max(x, f32(0.0)), f32(1.0))


def value(tex, u, v, p):
    kind = tex[0]
    if kind == "solid":
        return np.array(tex[1], np.float32)
    if kind == "checker":
        inv_scale = tex[1]
        with np.errstate(invalid="ignore"):
            lanes = [f32(np.floor(f32(inv_scale * c))) for c in (p[0], p[1], p[2], f32(0.0))]
            total = f32(f32(f32(lanes[0] + lanes[1]) + lanes[2]) + lanes[3])
        return value(tex[2] if rust_as_i32(total) % 2 == 0 else tex[3], u, v, p)
    if kind == "image":
        img = tex[1]
        h, w = img.shape[:2]
        uu, vv = clamp01(f32(u)), f32(f32(1.0) - clamp01(f32(v)))
        i, j = rust_as_u32(f32(uu * f32(w - 1))), rust_as_u32(f32(vv * f32(h - 1)))
        px = img[j, i].astype(np.float32) / f32(255.0) if img.dtype == np.uint8 else img[j, i]
        return (px * px).astype(np.float32)
    if kind == "noise":
        tables, scale = tex[1], tex[2]
        with np.errstate(over="ignore"):
            s = np.sin(f32(f32(scale * p[2]) + f32(f32(10.0) * turbulance(tables, p, 7))))
        return f32(f32(1.0) + s) * np.array([0.5, 0.5, 0.5], np.float32)
    raise ValueError(kind)


def test_textures_against_an_independent_restatement():
    o = ob.OracleScene()
    rng = np.random.default_rng(4)
    img8 = rng.integers(0, 256, (7, 11, 3), dtype=np.uint8)
    imgf = rng.uniform(0, 2, (3, 5, 3)).astype(np.float32)
    a, b = (0.1, 0.2, 0.3), (0.9, 0.8, 0.7)
    h_a, h_b, h_img8, h_imgf = o.tex_solid(*a), o.tex_solid(*b), o.tex_image(img8), o.tex_image(imgf)
    h_ck = o.tex_checker(0.64, h_a, h_b)
    h_ck2 = o.tex_checker(2.5, h_ck, h_img8)
    h_noise = o.tex_noise(o.perlin((4, 3, 2, 1), 64), 1.7)
    tables = perlin_new([4, 3, 2, 1], 64)
    py = {"a": ("solid", a), "b": ("solid", b), "img8": ("image", img8), "imgf": ("image", imgf)}
    py["ck"] = ("checker", f32(f32(1.0) / f32(0.64)), py["a"], py["b"])
    py["ck2"] = ("checker", f32(f32(1.0) / f32(2.5)), py["ck"], py["img8"])
    py["noise"] = ("noise", tables, f32(1.7))
    handles = {"img8": h_img8, "imgf": h_imgf, "ck": h_ck, "ck2": h_ck2, "noise": h_noise}
    mats = {k: o.lambertian(h) for k, h in handles.items()}
    o.set_world(o.bvh([o.sphere((i, 0, 0), 0.3, m) for i, m in enumerate(mats.values())]))
    o.camera((0, 0, -10), 4, 4, yaw=90.0)
    n = 200
    pts = np.concatenate([rng.normal(size=(n - 40, 3)) * 4, np.round(rng.normal(size=(40, 3)) * 3)]).astype(np.float32)
    uv = rng.choice(np.array([-0.5, 0.0, 1e-9, 0.25, 0.5, 0.73, 0.999999, 1.0, 1.5, np.nan], np.float32), (n, 2))
    for name, mat in mats.items():
        got = o.texture_value(mat, uv, pts)
        want = np.array([value(py[name], uv[i, 0], uv[i, 1], pts[i]) for i in range(n)], np.float32)
        if name == "noise":
            assert np.abs(got - want).max() <= 1e-6, name
        else:
            assert np.array_equal(got, want), name
