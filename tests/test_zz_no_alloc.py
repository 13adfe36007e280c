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


def _both(scene, w, h, first_sample, n_samples, seed):
    """(image through the allocating entry point, image through the no-alloc one), each into a zeroed device buffer."""
    n = w * h * 16
    out = []
    for noalloc in (False, True):
        d = C.c_void_p()
        scene._check(scene.amd.rt_buffer_create(scene.ctx, n, C.byref(d)))
        scene._check(scene.amd.rt_buffer_zero(scene.ctx, d, n))
        if noalloc:
            scene.render_accumulate_device_noalloc(d, first_sample, n_samples, seed=seed)
        else:
            scene.render_accumulate_device(d, first_sample, n_samples, seed=seed)
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
    # a tree whose traversal stack outgrows the LDS levels: the overflow area inside the array, on the smaller grid
    deep = rt.Scene()
    rng = np.random.default_rng(5)
    mats = [deep.lambertian(deep.tex_solid(*rng.uniform(0.1, 0.9, 3))) for _ in range(4)]
    deep.set_world(deep.bvh([deep.sphere(rng.uniform(-1, 1, 3), float(rng.uniform(5, 6)), mats[i % 4]) for i in range(1024)]))
    deep.camera((0, 0, 0), 48, 48, max_depth=4, vfov=90.0, background=(1, 1, 1), yaw=90.0)
    want, got = _both(deep, 48, 48, 1, 2, seed=2)
    assert np.array_equal(want, got)


def test_noalloc_entry_point_refuses_what_its_array_cannot_hold():
    s = rt.Scene.builtin(6, 16, 16, max_depth=1025)
    with pytest.raises(rt.RenderError, match="1024"):
        s.render_accumulate_device_noalloc(s.device_accumulator, 1, 1)
    ok = rt.Scene.builtin(6, 16, 16, max_depth=1024)
    ok.render_accumulate_device_noalloc(ok.device_accumulator, 1, 1)
    ok.synchronize()
