"""The entry point that allocates nothing (rt_render_accumulate_device_noalloc) against rt_render_accumulate_device.

Its ray queues live in a fixed array of the library's device image instead of memory the context allocates; the
image must be the same bit for bit.  Two calls in a row on one context (the second on a smaller image: the array
keeps the first call's contents), an image larger than the array's one batch (cut into pixel chunks), a tree that
needs the traversal-stack overflow area inside that array, and the documented limit.
"""
import ctypes as C

import numpy as np
import pytest

import raytracing_amd as rt

pytestmark = pytest.mark.gpu


def _both(scene, w, h, first_sample, n_samples, seed, tail_starts=(-1, -1)):
    """(image through the allocating entry point, image through the no-alloc one), each into a zeroed device buffer.
    tail_starts: the bounce at which the tail kernel takes over, for the allocating call and for the no-alloc call
    (-1: the library's choice, which for images this small is the tail kernel from the first bounce or the second)."""
    n = w * h * 16
    out = []
    for noalloc in (False, True):
        d = C.c_void_p()
        scene._check(scene.amd.rt_buffer_create(scene.ctx, n, C.byref(d)))
        scene._check(scene.amd.rt_buffer_zero(scene.ctx, d, n))
        scene.set_tail_start(tail_starts[1 if noalloc else 0])
        if noalloc:
            scene.render_accumulate_device_noalloc(d, first_sample, n_samples, seed=seed)
        else:
            scene.render_accumulate_device(d, first_sample, n_samples, seed=seed)
        scene.set_tail_start(-1)
        img = np.empty((h, w, 4), np.float32)
        scene._check(scene.amd.rt_buffer_download(scene.ctx, img.ctypes.data, d, n))
        scene._check(scene.amd.rt_buffer_destroy(scene.ctx, d))
        out.append(img)
    return out


def test_noalloc_entry_point_matches_the_allocating_one():
    big = rt.Scene.builtin(8, 96, 64)          # media, textures, placed leaves: every kernel of the path
    want, got = _both(big, 96, 64, 1, 8, seed=3)
    assert np.isfinite(got).all() and got[..., :3].max() > 0
    assert np.array_equal(want, got)
    small = rt.Scene.builtin(8, 40, 30)        # second call, smaller input: the array still holds the first call's queues
    want, got = _both(small, 40, 30, 5, 2, seed=3)
    assert np.array_equal(want, got)
    want, got = _both(big, 96, 64, 9, 3, seed=4)   # and back on the first context
    assert np.array_equal(want, got)


def test_noalloc_entry_point_chunks_large_images_and_deep_trees():
    s = rt.Scene.builtin(0, 640, 480)          # 307200 pixels: more than one batch of the fixed array
    want, got = _both(s, 640, 480, 1, 2, seed=1)
    assert np.array_equal(want, got)
    # a tree whose traversal stack outgrows the LDS levels, under the library's own schedule (the tail kernel's overflow area)
    deep = _deep_scene()
    want, got = _both(deep, 48, 48, 1, 2, seed=2)
    assert np.array_equal(want, got)


NEVER = 1 << 20   # a first tail bounce beyond any max_depth: every bounce runs the per-bounce kernels


def _deep_scene():
    """A tree whose traversal stack outgrows the levels the extend kernel keeps in LDS."""
    deep = rt.Scene()
    rng = np.random.default_rng(5)
    mats = [deep.lambertian(deep.tex_solid(*rng.uniform(0.1, 0.9, 3))) for _ in range(4)]
    deep.set_world(deep.bvh([deep.sphere(rng.uniform(-1, 1, 3), float(rng.uniform(5, 6)), mats[i % 4]) for i in range(1024)]))
    deep.camera((0, 0, 0), 48, 48, max_depth=4, vfov=90.0, background=(1, 1, 1), yaw=90.0)
    return deep


@pytest.mark.parametrize("schedules", [(0, NEVER), (NEVER, 0), (NEVER, NEVER)], ids=["tail_vs_per_bounce", "per_bounce_vs_tail", "per_bounce_both"])
def test_noalloc_per_bounce_kernels_on_the_fixed_array(schedules):
    """With the tail kernel switched off on the no-alloc side, ray generation, first leaves, extend (its overflow area inside
    the array, on the clamped grid), shade, textured shade (its list and counters behind the queue counters) and accumulate
    all run on the fixed layout; the allocating side runs the other schedule, so the two images come from different kernels."""
    big = rt.Scene.builtin(8, 96, 64)
    want, got = _both(big, 96, 64, 1, 8, seed=3, tail_starts=schedules)
    assert np.isfinite(got).all() and got[..., :3].max() > 0
    assert np.array_equal(want, got)
    textured = rt.Scene.builtin(100, 64, 48)   # earth + perlin spheres: every hit goes through the textured-shade list
    want, got = _both(textured, 64, 48, 1, 4, seed=5, tail_starts=schedules)
    assert np.array_equal(want, got)
    deep = _deep_scene()
    want, got = _both(deep, 48, 48, 1, 2, seed=2, tail_starts=schedules)
    assert np.array_equal(want, got)


def test_noalloc_entry_point_against_the_oracle_fixture():
    """The committed oracle image of the Book-2 final scene (tests/golden/scene8.npz, shared random numbers) through the
    no-alloc entry point, per-bounce kernels and tail kernel: the tolerances of the golden-image test of the allocating path."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene8.npz"))
    w, h, depth, spp, seed = int(g["width"]), int(g["height"]), int(g["max_depth"]), int(g["spp"]), int(g["seed"])
    want = g["acc_counter"]
    enc = lambda acc: np.sqrt(np.clip(acc[..., :3] / spp, 0.0, 1.0))
    for tail_start in (NEVER, -1):
        scene = rt.Scene.builtin(8, w, h, max_depth=depth)
        _, acc = _both(scene, w, h, 1, spp, seed=seed, tail_starts=(-1, tail_start))
        assert np.isfinite(acc).all() and (acc[..., 3] == 0).all()
        rmse = np.sqrt(np.mean((enc(acc) - enc(want)) ** 2))
        assert rmse <= 1e-3, rmse
        close = np.all(np.abs(acc[..., :3] - want[..., :3]) <= 1e-4 * np.maximum(1.0, np.abs(want[..., :3])), axis=-1)
        assert close.mean() >= 0.99, close.mean()


def test_noalloc_entry_point_allocates_nothing():
    """Free device memory as the runtime reports it does not move across calls of the entry point (after one warm-up call,
    so that loading the kernels' code objects is not counted), with both schedules, while the allocating entry point on a
    fresh context does take memory for its queues."""
    s = rt.Scene.builtin(8, 320, 200)
    d = C.c_void_p()
    s._check(s.amd.rt_buffer_create(s.ctx, 320 * 200 * 16, C.byref(d)))
    s._check(s.amd.rt_buffer_zero(s.ctx, d, 320 * 200 * 16))
    for tail_start in (NEVER, -1):
        s.set_tail_start(tail_start)
        s.render_accumulate_device_noalloc(d, 1, 2, seed=1)   # warm-up
        s.synchronize()
        free0, _ = s.memory_info()
        for k in range(3):
            s.render_accumulate_device_noalloc(d, 3 + 4 * k, 4, seed=1)
        s.synchronize()
        free1, _ = s.memory_info()
        assert free1 == free0, (tail_start, free0, free1)
    s.set_tail_start(NEVER)
    s.render_accumulate_device(d, 1, 4, seed=1)
    s.synchronize()
    free2, _ = s.memory_info()
    assert free2 < free1, "the allocating entry point should have taken queue memory (is the check measuring anything?)"
    s._check(s.amd.rt_buffer_destroy(s.ctx, d))


def test_noalloc_entry_point_refuses_what_its_array_cannot_hold():
    s = rt.Scene.builtin(6, 16, 16, max_depth=1025)
    with pytest.raises(rt.RenderError, match="1024"):
        s.render_accumulate_device_noalloc(s.device_accumulator, 1, 1)
    ok = rt.Scene.builtin(6, 16, 16, max_depth=1024)
    ok.render_accumulate_device_noalloc(ok.device_accumulator, 1, 1)
    ok.synchronize()
