"""The C ABI: header, shared library and Python binding agree (no GPU compute)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from raytracing_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_abi.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 25
    lib = C.CDLL(_abi.LIB_AMD_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in rt_abi.h but not exported by librt_amd.so"
    assert sorted(_abi.RT_ABI) == names, "raytracing_amd/_abi.py binds a different set of functions than the header declares"
    out = subprocess.run(["nm", "-D", "--defined-only", _abi.LIB_AMD_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (rt_[a-z0-9_]+)", out)))
    assert exported == names, "librt_amd.so exports rt_* symbols that the header does not declare"


def test_header_is_plain_c_and_layouts_match(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rt_abi.h"\nint main(void){\n'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(rt_ray), sizeof(rt_hit), sizeof(rt_camera_desc),'
                   ' sizeof(rt_counters), sizeof(rt_sphere), sizeof(rt_moving_sphere), sizeof(rt_quad), sizeof(rt_medium),'
                   ' sizeof(rt_material), sizeof(rt_texture), sizeof(rt_transform));\n'
                   'printf("%zu %zu %zu %zu\\n", offsetof(rt_ray, direction), offsetof(rt_ray, rng_pixel), offsetof(rt_hit, prim_id),'
                   ' offsetof(rt_camera_desc, background_colour));\nreturn 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    sizes, offsets = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    sizes = [int(x) for x in sizes.split()]
    assert sizes[:4] == [C.sizeof(_abi.RtRay), C.sizeof(_abi.RtHit), C.sizeof(_abi.RtCameraDesc), C.sizeof(_abi.RtCounters)]
    assert sizes == [48, 64, 104, 64, 32, 48, 48, 32, 16, 48, 32]
    assert [int(x) for x in offsets.split()] == [_abi.RtRay.direction.offset, _abi.RtRay.rng_pixel.offset, _abi.RtHit.prim_id.offset,
                                                   _abi.RtCameraDesc.background_colour.offset]
    assert _abi.ray_dtype().itemsize == 48 and _abi.hit_dtype().itemsize == 64 and _abi.leaf_dtype().itemsize == 88


def test_host_library_exports_its_table():
    lib = C.CDLL(_abi.LIB_HOST_PATH)
    for n in _abi.RTH_API:
        assert hasattr(lib, n), n


def test_version_and_argument_checks_without_a_gpu():
    amd = _abi.lib_amd()
    assert amd.rt_abi_version() == 1
    assert amd.rt_ctx_create(0, None) == -1  # RT_ERR_INVALID_ARGUMENT
    assert b"NULL" in amd.rt_last_error()
    assert amd.rt_ctx_destroy(None) == 0 and amd.rt_scene_destroy(None) == 0
    assert amd.rt_ctx_set_counters(None, 1) == -1
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        ctx = C.c_void_p()
        rc = amd.rt_ctx_create(0, C.byref(ctx))
        assert rc == -2 and not ctx.value  # RT_ERR_NO_DEVICE: the render path has no CPU fallback
        assert b"no HIP device" in amd.rt_last_error()


def test_product_does_not_reach_into_the_oracle():
    """Nothing under raytracing_amd/, bench.py's GPU leg or include/ may mention the oracle."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "raytracing_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle_binding|liboracle|oracle/|orc_", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
