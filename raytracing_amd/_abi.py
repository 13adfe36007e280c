"""ctypes bindings of the two shared libraries of this package.

``librt_amd.so``  — the gfx950 kernels behind the C ABI of ``include/rt_abi.h``.
``librt_host.so`` — the host-side scene API (C++ mirror of the reference crate's
                    construction API) exposed as ``rth_*`` C functions.

Loading fails loudly (ImportError) when a library has not been built: there is
no CPU fallback for the render path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
# RT_AMD_LIBRARY: another build of the same ABI (kernel tuning experiments)
LIB_AMD_PATH = os.environ.get("RT_AMD_LIBRARY") or os.path.join(_HERE, "csrc", "librt_amd.so")
LIB_HOST_PATH = os.path.join(_HERE, "host", "librt_host.so")
EARTH_ASSET = os.path.join(_HERE, "assets", "earthmap_1024x512.rgb8")
EARTH_SIZE = (1024, 512)

RT_NO_TRANSFORM = 0xFFFFFFFF
PRIM_SPHERE, PRIM_MOVING_SPHERE, PRIM_QUAD, PRIM_MEDIUM = 0, 1, 2, 3
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = 0, 1, 2, 3, 4
FINALIZE_MEAN, FINALIZE_SQRT, FINALIZE_TONEMAP_RGBA = 0, 1, 2
SPLIT_SAMPLES, SPLIT_ROWS = 0, 1


class RtRay(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("time", C.c_float), ("direction", C.c_float * 3), ("tmin", C.c_float),
                ("tmax", C.c_float), ("rng_pixel", C.c_uint32), ("rng_sample", C.c_uint32), ("rng_bounce", C.c_uint32)]


class RtHit(C.Structure):
    _fields_ = [("t", C.c_float), ("point", C.c_float * 3), ("normal", C.c_float * 3), ("u", C.c_float), ("v", C.c_float),
                ("prim_id", C.c_uint32), ("material", C.c_uint32), ("front_face", C.c_uint32), ("hit", C.c_uint32),
                ("_pad", C.c_uint32 * 3)]


class RtCameraDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_uint32), ("defocus_angle", C.c_float),
                ("centre", C.c_float * 3), ("pixel00_loc", C.c_float * 3), ("pixel_delta_u", C.c_float * 3),
                ("pixel_delta_v", C.c_float * 3), ("defocus_disk_u", C.c_float * 3), ("defocus_disk_v", C.c_float * 3),
                ("background_colour", C.c_float * 3), ("_pad", C.c_float)]


class RtCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("node_visits", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("quad_tests", C.c_uint64), ("medium_tests", C.c_uint64), ("hits", C.c_uint64), ("max_stack", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if not k.startswith("_")}


class RtSceneInfo(C.Structure):
    _fields_ = [("bvh_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("leaves_in_tree", C.c_uint32),
                ("leaves_tested_first", C.c_uint32), ("device_bytes", C.c_uint64), ("build_seconds", C.c_double)]


class RtBvhReport(C.Structure):
    _fields_ = [("nodes", C.c_uint32), ("depth", C.c_uint32), ("width", C.c_uint32), ("leaves_found", C.c_uint32),
                ("violations", C.c_uint32), ("sah_cost", C.c_float), ("stack_need", C.c_uint32), ("packed_surface_ratio", C.c_float)]


class Leaf(C.Structure):
    """One leaf of a flattened scene (object-space data plus its placement)."""
    _fields_ = [("kind", C.c_uint32), ("prim_id", C.c_uint32), ("material", C.c_uint32), ("medium", C.c_uint32),
                ("data", C.c_float * 12), ("sin_theta", C.c_float), ("cos_theta", C.c_float), ("offset", C.c_float * 3),
                ("has_transform", C.c_uint32)]


# numpy views of the same records
def ray_dtype():
    import numpy as np
    return np.dtype([("origin", np.float32, 3), ("time", np.float32), ("direction", np.float32, 3), ("tmin", np.float32),
                     ("tmax", np.float32), ("rng_pixel", np.uint32), ("rng_sample", np.uint32), ("rng_bounce", np.uint32)])


def hit_dtype():
    import numpy as np
    return np.dtype([("t", np.float32), ("point", np.float32, 3), ("normal", np.float32, 3), ("u", np.float32), ("v", np.float32),
                     ("prim_id", np.uint32), ("material", np.uint32), ("front_face", np.uint32), ("hit", np.uint32),
                     ("_pad", np.uint32, 3)])


def leaf_dtype():
    import numpy as np
    return np.dtype([("kind", np.uint32), ("prim_id", np.uint32), ("material", np.uint32), ("medium", np.uint32),
                     ("data", np.float32, 12), ("sin_theta", np.float32), ("cos_theta", np.float32), ("offset", np.float32, 3),
                     ("has_transform", np.uint32)])


_P = C.c_void_p
_U32 = C.c_uint32
_U64 = C.c_uint64
_F = C.c_float
_FP = C.POINTER(C.c_float)

# name -> (restype, argtypes); exactly the symbols include/rt_abi.h declares
RT_ABI = {
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_version": (_U32, []),
    "rt_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "rt_ctx_destroy": (C.c_int, [_P]),
    "rt_ctx_set_stream": (C.c_int, [_P, _P]),
    "rt_ctx_set_batch_paths": (C.c_int, [_P, _U64]),
    "rt_ctx_set_pipelines": (C.c_int, [_P, C.c_int]),
    "rt_ctx_set_tail_start": (C.c_int, [_P, C.c_int]),
    "rt_ctx_set_counters": (C.c_int, [_P, C.c_int]),
    "rt_ctx_get_counters": (C.c_int, [_P, C.POINTER(RtCounters), C.c_int]),
    "rt_buffer_create": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "rt_buffer_destroy": (C.c_int, [_P, _P]),
    "rt_buffer_zero": (C.c_int, [_P, _P, C.c_size_t]),
    "rt_buffer_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "rt_buffer_download": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "rt_host_buffer_create": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "rt_host_buffer_destroy": (C.c_int, [_P, _P]),
    "rt_scene_create": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "rt_scene_destroy": (C.c_int, [_P]),
    "rt_scene_get_info": (C.c_int, [_P, C.POINTER(RtSceneInfo)]),
    "rt_bvh_build_check": (C.c_int, [_FP, _U32, C.POINTER(RtBvhReport)]),
    "rt_render_accumulate": (C.c_int, [_P, C.POINTER(RtCameraDesc), _U64, _U32, _U32, _P]),
    "rt_ctx_set_host_buffer_pinning": (C.c_int, [_P, C.c_int]),
    "rt_ctx_release_host_buffer": (C.c_int, [_P]),
    "rt_ctx_memory_info": (C.c_int, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rt_render_accumulate_device": (C.c_int, [_P, C.POINTER(RtCameraDesc), _U64, _U32, _U32, _P]),
    "rt_render_accumulate_rows_device": (C.c_int, [_P, C.POINTER(RtCameraDesc), _U64, _U32, _U32, _U32, _U32, _P]),
    "rt_render_accumulate_multi": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.POINTER(RtCameraDesc), _U64, _U32, _U32, _P]),
    "rt_render_accumulate_device_noalloc": (C.c_int, [_P, C.POINTER(RtCameraDesc), _U64, _U32, _U32, _P]),
    "rt_finalize_device": (C.c_int, [_P, _P, _U64, _U32, C.c_int, _F, _P]),
    "rt_finalize": (C.c_int, [_P, _P, _U64, _U32, C.c_int, _F, _P]),
    "rt_intersect_batch": (C.c_int, [_P, _P, C.c_size_t, _U64, _P]),
    "rt_extend_batch": (C.c_int, [_P, _P, C.c_size_t, _U64, _P]),
    "rt_shade_batch": (C.c_int, [_P, _P, _P, C.c_size_t, _U64, _P, _P, _P]),
    "rt_primary_rays": (C.c_int, [_P, C.POINTER(RtCameraDesc), _U64, _U32, _P, C.c_size_t, _P]),
    "rt_ctx_synchronize": (C.c_int, [_P]),
    "rt_ctx_last_render_ms": (C.c_int, [_P, _FP]),
    "rt_ctx_set_phase_timing": (C.c_int, [_P, C.c_int]),
    "rt_ctx_phase_ms": (C.c_int, [_P, C.c_int, _FP, C.POINTER(_U32)]),
}

RTH_API = {
    "rth_last_error": (C.c_char_p, []),
    "rth_scene_new": (_P, [C.c_int]),
    "rth_scene_free": (None, [_P]),
    "rth_scene_builtin": (_P, [C.c_int, _U32, _U32, _U32, _P, _U32, _U32, _U64, C.c_int]),
    "rth_tex_solid": (_U32, [_P, _F, _F, _F]),
    "rth_tex_checker": (_U32, [_P, _F, _U32, _U32]),
    "rth_tex_image_rgb8": (_U32, [_P, _P, _U32, _U32]),
    "rth_tex_image_rgb32f": (_U32, [_P, _P, _U32, _U32]),
    "rth_perlin": (_U32, [_P, C.POINTER(_U64), _U32]),
    "rth_tex_noise": (_U32, [_P, _U32, _F]),
    "rth_material": (_U32, [_P, _U32, _U32, _F]),
    "rth_sphere": (_U32, [_P, _FP, _F, _U32]),
    "rth_moving_sphere": (_U32, [_P, _FP, _FP, _F, _U32]),
    "rth_quad": (_U32, [_P, _FP, _FP, _FP, _U32]),
    "rth_box": (_U32, [_P, _FP, _FP, _U32]),
    "rth_bvh": (_U32, [_P, C.POINTER(_U32), _U32]),
    "rth_move": (_U32, [_P, _U32, _FP]),
    "rth_rotate_y": (_U32, [_P, _U32, _F]),
    "rth_medium": (_U32, [_P, _U32, _F, _U32]),
    "rth_set_world": (None, [_P, _U32]),
    "rth_camera": (C.c_int, [_P, _FP, _U32, _U32, _U32, _F, _FP, _F, _F, _FP, _F, _F]),
    "rth_set_seed": (C.c_int, [_P, _U64]),
    "rth_render_samples": (C.c_int, [_P, _U32]),
    "rth_empty_render": (C.c_int, [_P]),
    "rth_hdr_render": (C.c_int, [_P, _P]),
    "rth_realtime_render": (C.c_int, [_P, _P]),
    "rth_realtime_render_view": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "rth_download_accumulator": (C.c_int, [_P, _P]),
    "rth_render_to_image": (C.c_int, [_P, _U64, C.c_char_p, _P]),
    "rth_save_checkpoint": (C.c_int, [_P, C.c_char_p]),
    "rth_load_checkpoint": (C.c_int, [_P, C.c_char_p]),
    "rth_write_exr": (C.c_int, [C.c_char_p, _U32, _U32, _P]),
    "rth_write_pfm": (C.c_int, [C.c_char_p, _U32, _U32, _P]),
    "rth_change_pitch_yaw_by": (C.c_int, [_P, _F, _F]),
    "rth_move_by": (C.c_int, [_P, _FP]),
    "rth_set_exposure": (C.c_int, [_P, _F]),
    "rth_samples": (_U64, [_P]),
    "rth_camera_desc": (C.c_int, [_P, C.POINTER(RtCameraDesc)]),
    "rth_ctx": (_P, [_P]),
    "rth_scene": (_P, [_P]),
    "rth_device_accumulator": (_P, [_P]),
    "rth_flat_counts": (C.c_int, [_P, C.POINTER(_U32)]),
    "rth_export_leaves": (_U32, [_P, _P, _U32]),
    "rth_material_count": (_U32, [_P]),
    "rth_material_get": (C.c_int, [_P, _U32, C.POINTER(_U32), _FP, C.POINTER(_U32), _FP]),
    "rth_perlin_count": (_U32, [_P]),
    "rth_perlin_export": (_U32, [_P, _U32, _P, _P, _P, _P]),
    "rth_scene_desc": (C.c_int, [_P, _P]),
    "rth_seed_next": (None, [C.POINTER(_U64), _U32, C.POINTER(_U64)]),
    "rth_seed_next_f32": (None, [C.POINTER(_U64), _U32, _FP]),
    "rth_raytracing_camera": (None, [_U32, _U32, _U32, _F, _FP, _FP, _FP, _F, _F, _FP, C.POINTER(RtCameraDesc)]),
}


def _bind(lib, table):
    for name, (restype, argtypes) in table.items():
        fn = getattr(lib, name)  # AttributeError here = a symbol the headers promise is missing
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


_amd = None
_host = None


def lib_amd():
    """The GPU library.  Loading needs the HIP runtime but no GPU; calls do."""
    global _amd
    if _amd is None:
        if not os.path.exists(LIB_AMD_PATH):
            raise ImportError(f"{LIB_AMD_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or `make -C raytracing_amd/csrc`); there is no CPU fallback")
        _amd = _bind(C.CDLL(LIB_AMD_PATH, mode=C.RTLD_GLOBAL), RT_ABI)
    return _amd


def lib_host():
    global _host
    if _host is None:
        lib_amd()
        if not os.path.exists(LIB_HOST_PATH):
            raise ImportError(f"{LIB_HOST_PATH} is missing: run `make -C raytracing_amd/host`")
        _host = _bind(C.CDLL(LIB_HOST_PATH), RTH_API)
    return _host
