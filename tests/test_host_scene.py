"""Host-side scene API (librt_host.so) against the oracle's construction: no GPU needed.

The host library builds the reference's scenes through its own mirror of the
reference API and flattens them into the C ABI's arrays; the oracle builds the
same scenes as a pointer tree.  Leaf by leaf the two must hold identical floats.
"""
import numpy as np
import pytest

import oracle_binding as ob
import raytracing_amd as rt
from raytracing_amd import _abi

SCENES = [0, 1, 2, 3, 4, 5, 6, 7, 8, 100, 101, 102]


def _sorted(leaves):
    order = np.lexsort((leaves["prim_id"], leaves["medium"]))
    return leaves[order]


@pytest.mark.parametrize("sid", SCENES)
def test_flattened_scene_equals_oracle_tree(sid):
    h = rt.Scene.builtin(sid, 64, 48, max_depth=13, stress_count=2000)
    o = ob.OracleScene.builtin(sid, 64, 48, max_depth=13, stress_count=2000)
    a, b = _sorted(h.leaves()), _sorted(o.leaves())
    assert len(a) == len(b)
    for f in ("kind", "prim_id", "material", "medium", "has_transform"):
        assert np.array_equal(a[f], b[f]), f
    geom = a["kind"] != _abi.PRIM_MEDIUM
    assert np.array_equal(a["data"][geom], b["data"][geom])
    assert np.array_equal(a["data"][~geom][:, 0], b["data"][~geom][:, 0])  # neg_inv_density
    for f in ("sin_theta", "cos_theta", "offset"):
        assert np.array_equal(a[f], b[f]), f
    assert h.materials() == o.materials()
    assert h.perlin_count == o.perlin_count
    for i in range(h.perlin_count):
        for x, y in zip(h.perlin_tables(i), o.perlin_tables(i)):
            assert np.array_equal(x, y)
    assert bytes(h.camera_desc()) == bytes(o.camera_desc())


def test_scene8_layout():
    h = rt.Scene.builtin(8, 32, 32)
    c = h.flat_counts()
    assert (c["spheres"], c["moving_spheres"], c["quads"], c["media"]) == (1006, 1, 2401, 2)
    assert c["world"] == 3410 and c["boundary"] == 2 and c["transforms"] == 1 and c["materials"] == 13
    assert c["images"] == 1 and c["perlins"] == 1
    d = h.camera_desc()
    assert (d.width, d.height, d.max_depth) == (32, 32, 25)
    assert tuple(d.centre) == (478.0, 278.0, -600.0) and tuple(d.background_colour) == (0.0, 0.0, 0.0)


def test_builder_api_matches_oracle_on_nested_placements():
    def build(s):
        t = s.tex_solid(0.2, 0.4, 0.6)
        m = s.lambertian(t)
        g = s.metal(s.tex_checker(0.5, s.tex_solid(0, 0, 0), s.tex_solid(1, 1, 1)), 0.3)
        inner = s.move(s.rotate_y(s.box((0, 0, 0), (1, 2, 3), m), 30.0), (5, 0, 1))
        outer = s.move(s.rotate_y(s.bvh([inner, s.sphere((0, 1, 0), 0.5, g)]), -45.0), (-2, 3, 4))
        fog = s.medium(s.move(s.moving_sphere((0, 0, 0), (0, 1, 0), 1.0, m), (7, 7, 7)), 0.5, t)
        s.set_world(s.bvh([outer, fog, s.quad((0, 0, 0), (1, 0, 0), (0, 1, 0), m)]))
        s.camera((0, 1, -10), 40, 30, max_depth=7, vfov=35.0, vup=(0, 1, 0), defocus_angle=0.6, focus_dist=9.0,
                 background=(0.1, 0.2, 0.3), pitch=-5.0, yaw=80.0)
        return s

    h, o = build(rt.Scene()), build(ob.OracleScene())
    a, b = _sorted(h.leaves()), _sorted(o.leaves())
    assert len(a) == len(b) == 6 + 1 + 1 + 1 + 1
    for f in ("kind", "prim_id", "material", "medium", "has_transform"):
        assert np.array_equal(a[f], b[f]), f
    # composed Move/RotateY chains agree to rounding (the composition order of the float ops is the same)
    for f in ("sin_theta", "cos_theta", "offset"):
        assert np.array_equal(a[f], b[f]), f
    assert bytes(h.camera_desc()) == bytes(o.camera_desc())
    assert h.camera_desc().defocus_angle == pytest.approx(0.6)


def test_camera_state_machine():
    h = rt.Scene.builtin(5, 32, 20)
    d0 = bytes(h.camera_desc())
    assert h.samples == 0
    h.change_pitch_yaw_by(0.0, 0.0)  # no-op (src/camera.rs:150)
    assert bytes(h.camera_desc()) == d0
    h.change_pitch_yaw_by(10.0, 20.0)
    d1 = h.camera_desc()
    assert bytes(d1) != d0
    h.move_by((1.0, 2.0, 3.0))
    d2 = h.camera_desc()
    assert tuple(d2.centre) == (-24.0, 2.0, 3.0)
    o = ob.OracleScene.builtin(5, 32, 20)
    o.lib.orc_camera  # the oracle exposes the same pitch/yaw arithmetic through its builder call
    ref = ob.OracleScene()
    t = ref.tex_solid(0, 0, 0)
    ref.set_world(ref.bvh([ref.sphere((0, 0, 0), 1, ref.lambertian(t))]))
    ref.camera((-24.0, 2.0, 3.0), 32, 20, max_depth=25, vfov=20.0, vup=(0, 2, 0), defocus_angle=0.0, focus_dist=10.0,
               background=(0.7, 0.7, 0.7), pitch=10.0, yaw=20.0)
    assert bytes(ref.camera_desc()) == bytes(d2)


def test_seed_and_errors():
    import ctypes as C
    host = _abi.lib_host()
    st = (C.c_uint64 * 4)(69, 420, 420, 69)
    out = (C.c_uint64 * 4)()
    host.rth_seed_next(st, 4, out)
    assert list(out) == [0x45000045, 0xD2001FB4, 0x03FE2000F0802441, 0x04822480EC481C21]
    with pytest.raises(rt.RenderError, match="0..=8"):
        rt.Scene.builtin(9)
    s = rt.Scene()
    with pytest.raises(rt.RenderError, match="no camera"):
        s.render_samples(1)
