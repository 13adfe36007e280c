"""The scene builders are data: every numeric constant of the reference's builders (src/main.rs:291-731) must appear, as often,
in the two places that restate them - the host library's scenes (raytracing_amd/host/scenes.cpp) and the oracle's
(oracle/oracle_scenes.hpp).  The reference is read as text, where it is present; the differences that are allowed are the ones
listed below, each with its reason.  (This does not check which constant goes where - the scene census and the look-alike
renders do - but a mistyped or dropped number shows.)"""
import os
import re
from collections import Counter

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/src/main.rs"

pytestmark = pytest.mark.skipif(not os.path.exists(REFERENCE), reason="the reference's sources are not on this machine")


def bodies(text, pattern):
    out = {}
    for m in re.finditer(pattern, text, re.M):
        i = text.index("{", m.end() - 1)
        depth, j = 0, i
        while True:
            depth += {"{": 1, "}": -1}.get(text[j], 0)
            if depth == 0:
                break
            j += 1
        out[m.group(1)] = text[i:j + 1]
    return out


def literals(body):
    body = re.sub(r"//.*", "", body)
    number = r"(?<![\w.])(-?\d+(?:_\d+)*(?:\.\d*)?(?:[eE][-+]?\d+)?)(?:f32|f64|f|usize|u32|u64|isize|i32|u)?(?![\w.])"
    return Counter(float(m.group(1).replace("_", "")) for m in re.finditer(number, body))


# small whole numbers are indices, loop bounds and unit vectors on both sides
SMALL = {0.0, 1.0, 2.0, 3.0, 4.0}
# resolution: the reference computes (16 * resolution_scale, 9 * resolution_scale) or (10 * scale, 10 * scale); the restatements
# hold the default sizes 1600 x 900 and 1000 x 1000
SQUARE = (Counter({10.0: 2}), Counter({1000.0: 2}))
WIDE = (Counter({9.0: 1, 16.0: 1}), Counter({900.0: 1, 1600.0: 1}))
WIDE_ONE = (Counter({10.0: 1, 16.0: 1}), Counter({1000.0: 1, 1600.0: 1}))   # 16 * scale by 10 * scale
GLASS = Counter({1.5: 1, 90.0: 2, 190.0: 2})       # the optional glass sphere of the benchmark's Cornell box variant
GRID = Counter({-11.0: 2, 11.0: 2})                # `for a in -11..11` is not a number followed by a word boundary in Rust

HOST = {
    "the_final_scene": (["the_final_scene"], SQUARE[0], SQUARE[1]),
    "cornell_box": (["cornell_box", "cornell_walls", "cornell_camera"], SQUARE[0], SQUARE[1] + GLASS),
    "cornell_box_fog": (["cornell_box_fog", "cornell_walls", "cornell_camera"], SQUARE[0], SQUARE[1]),
    "simple_light": (["simple_light"], WIDE[0], WIDE[1]),
    "quads": (["quads"], SQUARE[0], SQUARE[1]),
    # one function builds the day and the night scene: the other one's background, lights and checker scale are extra
    "original_bouncing_spheres": (["bouncing_spheres", "checker"], WIDE[0], WIDE[1] + GRID + Counter({0.125: 1, 0.5: 1, 0.8: 1, 25.0: 1, 64.0: 1})),
    "bouncing_spheres_night": (["bouncing_spheres", "checker"], WIDE[0], WIDE[1] + GRID + Counter({0.7: 3, 0.8: 1})),
    "world_sphere": (["world_sphere"], WIDE_ONE[0], WIDE_ONE[1]),
    "checkered_spheres": (["checkered_spheres", "checker"], WIDE_ONE[0], WIDE_ONE[1]),
}
ORACLE = dict(HOST)
ORACLE.update({
    "the_final_scene": (["the_final_scene"], SQUARE[0], SQUARE[1] + Counter({165.0: 2})),   # box2's size written out once more
    "original_bouncing_spheres": (["original_bouncing_spheres"], WIDE[0], WIDE[1] + GRID + Counter({0.5: 2})),
    "bouncing_spheres_night": (["bouncing_spheres_night"], WIDE[0], WIDE[1] + GRID + Counter({0.5: 2})),
    "checkered_spheres": (["checkered_spheres"], WIDE_ONE[0], WIDE_ONE[1]),
})


def strip(counter):
    return Counter({k: v for k, v in counter.items() if k not in SMALL})


@pytest.mark.parametrize("which", ["host", "oracle"])
def test_every_constant_of_the_reference_scene_builders_is_restated(which):
    ref = bodies(open(REFERENCE).read(), r"^fn (\w+)<'a>\(.*\{\s*$")
    if which == "host":
        mine = bodies(open(os.path.join(ROOT, "raytracing_amd", "host", "scenes.cpp")).read(), r"^[\w:<>, ]+ (\w+)\(.*\) \{\s*$")
        table = HOST
    else:
        mine = bodies(open(os.path.join(ROOT, "oracle", "oracle_scenes.hpp")).read(), r"^inline \w+ (\w+)\(.*\) \{\s*$")
        table = ORACLE
    assert set(table) <= set(ref)
    for scene, (functions, only_reference, only_here) in table.items():
        theirs = literals(ref[scene])
        ours = Counter()
        for f in functions:
            ours += literals(mine[f])
        assert len(theirs) >= 8, scene
        assert strip(theirs - ours) == strip(only_reference), (scene, "missing here", dict(strip(theirs - ours)))
        assert strip(ours - theirs) == strip(only_here), (scene, "not in the reference", dict(strip(ours - theirs)))


def test_every_constant_of_the_path_s_formulas_is_restated():
    """The thresholds and factors of the files on the path (0.001, 1e-8, 0.0001, 255.999, the rotation counts of the generator
    ...): each occurs in the oracle and in the kernels or the host library.  Not restated as written: the unused global SEED
    [6, 9, 4, 20] of src/rng.rs:7, and FRACTION_BITS = 52 (src/rng.rs:44), which both sides use as 52 + 1 = 53."""
    def read(*parts):
        return open(os.path.join(ROOT, *parts)).read()
    oracle = literals(read("oracle", "oracle_core.hpp")) + literals(read("oracle", "oracle_capi.cpp"))
    product = Counter()
    for parts in (("raytracing_amd", "csrc", "rt_kernels.hip"), ("raytracing_amd", "csrc", "rt_device.h"), ("raytracing_amd", "csrc", "rt_api.hip"),
                  ("raytracing_amd", "host", "raytracing.cpp"), ("raytracing_amd", "host", "raytracing.hpp")):
        product += literals(read(*parts))
    seen = 0
    for f in ("math/ray.rs", "hittable.rs", "material.rs", "texture.rs", "perlin_noise.rs", "camera.rs", "math/aabb.rs", "math/vec3.rs", "rng.rs"):
        text = open(os.path.join(os.path.dirname(REFERENCE), f)).read().split("#[cfg(test)]")[0]
        for value in strip(literals(text)):
            seen += 1
            if f == "rng.rs" and value in (20.0, 52.0):
                continue
            assert value in oracle, (f, value, "not in the oracle")
            assert value in product, (f, value, "not in the kernels or the host library")
    assert seen >= 25
