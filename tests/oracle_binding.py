"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module: the oracle is the checker, never the thing that ships.
"""
import ctypes as C
import os

import numpy as np

from raytracing_amd import _abi
from raytracing_amd.scene import BuilderCalls, load_earth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# RT_ORACLE_LIBRARY: another build of the same oracle (bench.py's -march=native build for the CPU baseline)
LIB_PATH = os.environ.get("RT_ORACLE_LIBRARY") or os.path.join(ROOT, "oracle", "liboracle.so")

RNG_XOSHIRO, RNG_COUNTER = 0, 1

_P, _U32, _U64, _F = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
_FP = C.POINTER(C.c_float)

_API = {
    "orc_scene_new": (_P, []),
    "orc_scene_free": (None, [_P]),
    "orc_scene_builtin": (_P, [C.c_int, _U32, _U32, _U32, _P, _U32, _U32, _U64]),
    "orc_tex_solid": (_U32, [_P, _F, _F, _F]),
    "orc_tex_checker": (_U32, [_P, _F, _U32, _U32]),
    "orc_tex_image_rgb8": (_U32, [_P, _P, _U32, _U32]),
    "orc_tex_image_rgb32f": (_U32, [_P, _P, _U32, _U32]),
    "orc_perlin": (_U32, [_P, C.POINTER(_U64), _U32]),
    "orc_tex_noise": (_U32, [_P, _U32, _F]),
    "orc_material": (_U32, [_P, _U32, _U32, _F]),
    "orc_sphere": (_U32, [_P, _FP, _F, _U32]),
    "orc_moving_sphere": (_U32, [_P, _FP, _FP, _F, _U32]),
    "orc_quad": (_U32, [_P, _FP, _FP, _FP, _U32]),
    "orc_box": (_U32, [_P, _FP, _FP, _U32]),
    "orc_bvh": (_U32, [_P, C.POINTER(_U32), _U32]),
    "orc_move": (_U32, [_P, _U32, _FP]),
    "orc_rotate_y": (_U32, [_P, _U32, _F]),
    "orc_medium": (_U32, [_P, _U32, _F, _U32]),
    "orc_set_world": (None, [_P, _U32]),
    "orc_camera": (None, [_P, _FP, _U32, _U32, _U32, _F, _FP, _F, _F, _FP, _F, _F]),
    "orc_camera_desc": (None, [_P, C.POINTER(_abi.RtCameraDesc)]),
    "orc_material_count": (_U32, [_P]),
    "orc_material_get": (None, [_P, _U32, C.POINTER(_U32), _FP, C.POINTER(_U32), _FP]),
    "orc_bvh_assert_failures": (_U32, [_P]),
    "orc_export_leaves": (_U32, [_P, _P, _U32]),
    "orc_perlin_count": (_U32, [_P]),
    "orc_perlin_export": (_U32, [_P, _U32, _P, _P, _P, _P]),
    "orc_render": (None, [_P, C.c_int, _U64, _U32, _U32, _P, C.c_int, C.POINTER(_abi.RtCounters)]),
    "orc_intersect": (None, [_P, _P, C.c_size_t, _U64, _P]),
    "orc_shade": (None, [_P, _P, _P, C.c_size_t, _U64, _P, _P, _P]),
    "orc_texture_value": (None, [_P, _U32, C.c_size_t, _P, _P, _P]),
    "orc_perlin_eval": (None, [_P, _U32, C.c_size_t, _P, _U32, _P, _P]),
    "orc_primary_rays": (None, [_P, _U64, _U32, _P, C.c_size_t, _P]),
    "orc_seed_next": (None, [C.POINTER(_U64), _U32, C.POINTER(_U64)]),
    "orc_seed_next_f32": (None, [C.POINTER(_U64), _U32, _FP]),
    "orc_counter_draws": (None, [_U64, _U32, _U32, _U32, _U32, _U32, C.POINTER(_U32), _FP]),
    "orc_finalize": (None, [_P, _U64, _U32, C.c_int, _F, _P]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C oracle`")
        _lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _API.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = restype, argtypes
    return _lib


class OracleScene(BuilderCalls):
    def __init__(self, handle=None):
        self.lib = lib()
        if handle is None:
            handle = self.lib.orc_scene_new()
        super().__init__(self.lib, "orc_", handle)

    @classmethod
    def builtin(cls, scene_id, width=0, height=0, max_depth=25, stress_count=100000):
        l = lib()
        earth = load_earth() if scene_id in (4, 8, 100) else None
        ptr = earth.ctypes.data if earth is not None else None
        ew, eh = _abi.EARTH_SIZE if earth is not None else (0, 0)
        h = l.orc_scene_builtin(scene_id, width, height, max_depth, ptr, ew, eh, stress_count)
        if not h:
            raise RuntimeError(f"oracle cannot build scene {scene_id}")
        return cls(handle=h)

    def __del__(self):
        try:
            if self._h:
                self.lib.orc_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    def camera_desc(self):
        d = _abi.RtCameraDesc()
        self.lib.orc_camera_desc(self._h, C.byref(d))
        return d

    @property
    def resolution(self):
        d = self.camera_desc()
        return int(d.width), int(d.height)

    def leaves(self):
        n = self.lib.orc_export_leaves(self._h, None, 0)
        out = np.zeros(n, _abi.leaf_dtype())
        self.lib.orc_export_leaves(self._h, out.ctypes.data, n)
        return out

    def materials(self):
        out = []
        for i in range(self.lib.orc_material_count(self._h)):
            kind, param, tk, col = C.c_uint32(), C.c_float(), C.c_uint32(), (C.c_float * 3)()
            self.lib.orc_material_get(self._h, i, C.byref(kind), C.byref(param), C.byref(tk), col)
            out.append((kind.value, param.value, tk.value, tuple(col)))
        return out

    @property
    def perlin_count(self):
        return int(self.lib.orc_perlin_count(self._h))

    def perlin_tables(self, idx):
        n = self.lib.orc_perlin_export(self._h, idx, None, None, None, None)
        vecs = np.zeros((n, 3), np.float32)
        px, py, pz = (np.zeros(n, np.uint32) for _ in range(3))
        self.lib.orc_perlin_export(self._h, idx, vecs.ctypes.data, px.ctypes.data, py.ctypes.data, pz.ctypes.data)
        return vecs, px, py, pz

    @property
    def bvh_assert_failures(self):
        return int(self.lib.orc_bvh_assert_failures(self._h))

    def render(self, n_samples, first_sample=1, mode=RNG_COUNTER, seed=0, threads=None, acc=None, counters=None):
        w, h = self.resolution
        if acc is None:
            acc = np.zeros((h, w, 4), np.float32)
        if threads is None:
            threads = min(16, os.cpu_count() or 1)
        cptr = C.byref(counters) if counters is not None else None
        self.lib.orc_render(self._h, mode, seed, first_sample, n_samples, acc.ctypes.data, threads, cptr)
        return acc

    def intersect(self, rays, seed=0):
        rays = np.ascontiguousarray(rays, dtype=_abi.ray_dtype())
        out = np.zeros(len(rays), _abi.hit_dtype())
        self.lib.orc_intersect(self._h, rays.ctypes.data, len(rays), seed, out.ctypes.data)
        return out

    def shade(self, rays, hits, seed=0):
        rays = np.ascontiguousarray(rays, dtype=_abi.ray_dtype())
        hits = np.ascontiguousarray(hits, dtype=_abi.hit_dtype())
        n = len(rays)
        sc = np.zeros(n, _abi.ray_dtype())
        att = np.zeros((n, 3), np.float32)
        emit = np.zeros((n, 3), np.float32)
        self.lib.orc_shade(self._h, rays.ctypes.data, hits.ctypes.data, n, seed, sc.ctypes.data, att.ctypes.data, emit.ctypes.data)
        return sc, att, emit

    def primary_rays(self, pixels, sample, seed=0):
        pixels = np.ascontiguousarray(pixels, dtype=np.uint32)
        out = np.zeros(len(pixels), _abi.ray_dtype())
        self.lib.orc_primary_rays(self._h, seed, sample, pixels.ctypes.data, len(pixels), out.ctypes.data)
        return out

    def texture_value(self, material, uv, p):
        uv = np.ascontiguousarray(uv, np.float32)
        p = np.ascontiguousarray(p, np.float32)
        out = np.zeros((len(p), 3), np.float32)
        self.lib.orc_texture_value(self._h, material, len(p), uv.ctypes.data, p.ctypes.data, out.ctypes.data)
        return out

    def perlin_eval(self, perlin, p, depth=7):
        p = np.ascontiguousarray(p, np.float32)
        noise = np.zeros(len(p), np.float32)
        turb = np.zeros(len(p), np.float32)
        self.lib.orc_perlin_eval(self._h, perlin, len(p), p.ctypes.data, depth, noise.ctypes.data, turb.ctypes.data)
        return noise, turb


def seed_next(state, n):
    st = (C.c_uint64 * 4)(*state)
    out = (C.c_uint64 * n)()
    lib().orc_seed_next(st, n, out)
    return [int(x) for x in out]


def seed_next_f32(state, n):
    st = (C.c_uint64 * 4)(*state)
    out = (C.c_float * n)()
    lib().orc_seed_next_f32(st, n, out)
    return [float(x) for x in out]


def counter_draws(seed, pixel, sample, bounce, stream, n):
    u = (C.c_uint32 * n)()
    f = (C.c_float * n)()
    lib().orc_counter_draws(seed, pixel, sample, bounce, stream, n, u, f)
    return np.array(u, dtype=np.uint32), np.array(f, dtype=np.float32)


def finalize(acc, total_samples, mode, exposure=1.0):
    acc = np.ascontiguousarray(acc, np.float32)
    n = acc.size // 4
    out = np.empty(acc.shape[:-1], np.uint32) if mode == 2 else np.empty(acc.shape, np.float32)
    lib().orc_finalize(acc.ctypes.data, n, total_samples, mode, exposure, out.ctypes.data)
    return out
