"""Python face of the host-side scene API (``librt_host.so``) and of the C ABI.

``Scene`` mirrors the reference crate's ``Camera`` + scene-construction calls:
build a scene (a built-in one, or leaf by leaf), then ``render_samples`` /
``hdr_render`` / ``realtime_render`` run the render loop on the GPU.  The per-ray
entry points of the C ABI (``intersect``, ``shade``, ``primary_rays``) are
exposed for parity checks.
"""
import ctypes as C

import numpy as np

from . import _abi


class RenderError(RuntimeError):
    pass


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def write_exr(path, rgba):
    """RGBA float32 image (H, W, 4) as an uncompressed scanline OpenEXR file."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    host = _abi.lib_host()
    if host.rth_write_exr(path.encode(), rgba.shape[1], rgba.shape[0], rgba.ctypes.data):
        raise RenderError(host.rth_last_error().decode())


def write_pfm(path, rgba):
    rgba = np.ascontiguousarray(rgba, np.float32)
    host = _abi.lib_host()
    if host.rth_write_pfm(path.encode(), rgba.shape[1], rgba.shape[0], rgba.ctypes.data):
        raise RenderError(host.rth_last_error().decode())


def load_earth():
    """The decoded earthmap.jpg (PIL 12.2 / libjpeg-turbo) as H x W x 3 uint8."""
    w, h = _abi.EARTH_SIZE
    return np.fromfile(_abi.EARTH_ASSET, dtype=np.uint8).reshape(h, w, 3)


class BuilderCalls:
    """Scene-construction calls shared by every library that exports the
    ``<prefix>tex_solid`` ... ``<prefix>camera`` C functions; returns integer handles."""

    def __init__(self, lib, prefix, handle):
        self._lib, self._prefix, self._h = lib, prefix, handle

    def _fn(self, name):
        return getattr(self._lib, self._prefix + name)

    # textures
    def tex_solid(self, r, g, b):
        return self._fn("tex_solid")(self._h, r, g, b)

    def tex_checker(self, scale, even, odd):
        return self._fn("tex_checker")(self._h, scale, even, odd)

    def tex_image(self, img):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        if img.dtype == np.uint8:
            return self._fn("tex_image_rgb8")(self._h, img.ctypes.data, w, h)
        img = np.ascontiguousarray(img, dtype=np.float32)
        return self._fn("tex_image_rgb32f")(self._h, img.ctypes.data, w, h)

    def perlin(self, seed4, point_count):
        return self._fn("perlin")(self._h, (C.c_uint64 * 4)(*seed4), point_count)

    def tex_noise(self, perlin, scale):
        return self._fn("tex_noise")(self._h, perlin, scale)

    # materials (return MaterialId)
    def material(self, kind, tex, param=0.0):
        return self._fn("material")(self._h, kind, tex, param)

    def lambertian(self, tex):
        return self.material(_abi.MAT_LAMBERTIAN, tex)

    def metal(self, tex, fuzz):
        return self.material(_abi.MAT_METAL, tex, fuzz)

    def dielectric(self, tex, ri):
        return self.material(_abi.MAT_DIELECTRIC, tex, ri)

    def diffuse_light(self, tex):
        return self.material(_abi.MAT_DIFFUSE_LIGHT, tex)

    def isotropic(self, tex):
        return self.material(_abi.MAT_ISOTROPIC, tex)

    # hittables
    def sphere(self, centre, radius, mat):
        return self._fn("sphere")(self._h, _f3(centre), radius, mat)

    def moving_sphere(self, c1, c2, radius, mat):
        return self._fn("moving_sphere")(self._h, _f3(c1), _f3(c2), radius, mat)

    def quad(self, q, u, v, mat):
        return self._fn("quad")(self._h, _f3(q), _f3(u), _f3(v), mat)

    def box(self, a, b, mat):
        return self._fn("box")(self._h, _f3(a), _f3(b), mat)

    def bvh(self, handles):
        arr = (C.c_uint32 * len(handles))(*handles)
        return self._fn("bvh")(self._h, arr, len(handles))

    def move(self, h, offset):
        return self._fn("move")(self._h, h, _f3(offset))

    def rotate_y(self, h, degrees):
        return self._fn("rotate_y")(self._h, h, degrees)

    def medium(self, boundary, density, tex):
        return self._fn("medium")(self._h, boundary, density, tex)

    def set_world(self, h):
        self._fn("set_world")(self._h, h)

    def camera(self, pos, width, height, max_depth=25, vfov=40.0, vup=(0, 1, 0), defocus_angle=0.0, focus_dist=10.0,
               background=(0, 0, 0), pitch=0.0, yaw=0.0):
        return self._fn("camera")(self._h, _f3(pos), width, height, max_depth, vfov, _f3(vup), defocus_angle, focus_dist,
                                  _f3(background), pitch, yaw)


class Scene(BuilderCalls):
    """A scene plus its Camera on one GPU."""

    def __init__(self, handle=None, device=0):
        self.host = _abi.lib_host()
        self.amd = _abi.lib_amd()
        if handle is None:
            handle = self.host.rth_scene_new(device)
        super().__init__(self.host, "rth_", handle)
        self._keep = []

    @classmethod
    def builtin(cls, scene_id, width=0, height=0, max_depth=25, device=0, stress_count=100000):
        """The reference's scene `scene_id` (0..8) or a benchmark configuration (100..102)."""
        host = _abi.lib_host()
        earth = load_earth() if scene_id in (4, 8, 100) else None
        ptr = earth.ctypes.data if earth is not None else None
        ew, eh = _abi.EARTH_SIZE if earth is not None else (0, 0)
        h = host.rth_scene_builtin(scene_id, width, height, max_depth, ptr, ew, eh, stress_count, device)
        if not h:
            raise RenderError(host.rth_last_error().decode())
        return cls(handle=h, device=device)

    def close(self):
        if self._h:
            self.host.rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_host(self, rc):
        if rc != 0:
            raise RenderError(self.host.rth_last_error().decode())

    def _check(self, rc):
        if rc != 0:
            raise RenderError(self.amd.rt_last_error().decode())

    def camera(self, *a, **k):
        self._check_host(super().camera(*a, **k))

    # ---- Camera operations (src/camera.rs:66-168) ----
    def set_seed(self, seed):
        self._check_host(self.host.rth_set_seed(self._h, seed))

    def render_samples(self, n):
        self._check_host(self.host.rth_render_samples(self._h, n))

    def empty_render(self):
        self._check_host(self.host.rth_empty_render(self._h))

    def hdr_render(self):
        w, h = self.resolution
        out = np.empty((h, w, 3), np.float32)
        self._check_host(self.host.rth_hdr_render(self._h, out.ctypes.data))
        return out

    def realtime_render(self, copy=True):
        """One more sample per pixel, tone-mapped and packed (0BGR words, src/camera.rs:89-104).  copy=False returns a view
        of the camera's own page-locked frame instead of a fresh array - what the reference's `&[u32]` is: valid until
        this scene's next render call, and 0.1 - 0.2 ms per frame cheaper at 800x800."""
        w, h = self.resolution
        if copy:
            out = np.empty((h, w), np.uint32)
            self._check_host(self.host.rth_realtime_render(self._h, out.ctypes.data))
            return out
        words, n = C.c_void_p(), C.c_uint64()
        self._check_host(self.host.rth_realtime_render_view(self._h, C.byref(words), C.byref(n)))
        frame = np.ctypeslib.as_array(C.cast(words, C.POINTER(C.c_uint32)), shape=(h, w))
        frame.flags.writeable = False
        return frame

    def accumulator(self):
        w, h = self.resolution
        out = np.empty((h, w, 4), np.float32)
        self._check_host(self.host.rth_download_accumulator(self._h, out.ctypes.data))
        return out

    def render_to_image(self, sample_count, path=None):
        """`--image` of the reference: sample_count - 1 samples, then hdr_render; RGBA float32 (alpha 1),
        written to `path` as OpenEXR (or PFM for *.pfm) when given."""
        w, h = self.resolution
        out = np.empty((h, w, 4), np.float32)
        self._check_host(self.host.rth_render_to_image(self._h, sample_count, path.encode() if path else None, out.ctypes.data))
        return out

    def save_checkpoint(self, path):
        self._check_host(self.host.rth_save_checkpoint(self._h, path.encode()))

    def load_checkpoint(self, path):
        self._check_host(self.host.rth_load_checkpoint(self._h, path.encode()))

    def change_pitch_yaw_by(self, dp, dy):
        self._check_host(self.host.rth_change_pitch_yaw_by(self._h, dp, dy))

    def move_by(self, step):
        self._check_host(self.host.rth_move_by(self._h, _f3(step)))

    def set_exposure(self, e):
        self._check_host(self.host.rth_set_exposure(self._h, e))

    @property
    def samples(self):
        return int(self.host.rth_samples(self._h))

    def camera_desc(self):
        d = _abi.RtCameraDesc()
        self._check_host(self.host.rth_camera_desc(self._h, C.byref(d)))
        return d

    @property
    def resolution(self):
        d = self.camera_desc()
        return int(d.width), int(d.height)

    # ---- flattened scene (host only, no GPU needed) ----
    def flat_counts(self):
        out = (C.c_uint32 * 11)()
        self._check_host(self.host.rth_flat_counts(self._h, out))
        names = ["spheres", "moving_spheres", "quads", "media", "world", "boundary", "transforms", "materials", "textures",
                 "images", "perlins"]
        return dict(zip(names, [int(x) for x in out]))

    def leaves(self):
        n = self.host.rth_export_leaves(self._h, None, 0)
        out = np.zeros(n, _abi.leaf_dtype())
        self.host.rth_export_leaves(self._h, out.ctypes.data, n)
        return out

    def materials(self):
        out = []
        for i in range(self.host.rth_material_count(self._h)):
            kind, param, tk, col = C.c_uint32(), C.c_float(), C.c_uint32(), (C.c_float * 3)()
            self._check_host(self.host.rth_material_get(self._h, i, C.byref(kind), C.byref(param), C.byref(tk), col))
            out.append((kind.value, param.value, tk.value, tuple(col)))
        return out

    def perlin_tables(self, idx):
        n = self.host.rth_perlin_export(self._h, idx, None, None, None, None)
        vecs = np.zeros((n, 3), np.float32)
        px, py, pz = (np.zeros(n, np.uint32) for _ in range(3))
        self.host.rth_perlin_export(self._h, idx, vecs.ctypes.data, px.ctypes.data, py.ctypes.data, pz.ctypes.data)
        return vecs, px, py, pz

    @property
    def perlin_count(self):
        return int(self.host.rth_perlin_count(self._h))

    # ---- C ABI handles ----
    @property
    def ctx(self):
        p = self.host.rth_ctx(self._h)
        if not p:
            raise RenderError(self.host.rth_last_error().decode())
        return p

    @property
    def rt_scene(self):
        p = self.host.rth_scene(self._h)
        if not p:
            raise RenderError(self.host.rth_last_error().decode())
        return p

    @property
    def device_accumulator(self):
        p = self.host.rth_device_accumulator(self._h)
        if not p:
            raise RenderError(self.host.rth_last_error().decode())
        return p

    # ---- per-ray entry points of the C ABI ----
    def intersect(self, rays, seed=0):
        rays = np.ascontiguousarray(rays, dtype=_abi.ray_dtype())
        out = np.zeros(len(rays), _abi.hit_dtype())
        self._check(self.amd.rt_intersect_batch(self.rt_scene, rays.ctypes.data, len(rays), seed, out.ctypes.data))
        return out

    def extend(self, rays, seed=0):
        """The same closest hits as `intersect`, computed by the render path's own traversal kernels."""
        rays = np.ascontiguousarray(rays, dtype=_abi.ray_dtype())
        out = np.zeros(len(rays), _abi.hit_dtype())
        self._check(self.amd.rt_extend_batch(self.rt_scene, rays.ctypes.data, len(rays), seed, out.ctypes.data))
        return out

    def shade(self, rays, hits, seed=0):
        rays = np.ascontiguousarray(rays, dtype=_abi.ray_dtype())
        hits = np.ascontiguousarray(hits, dtype=_abi.hit_dtype())
        n = len(rays)
        sc = np.zeros(n, _abi.ray_dtype())
        att = np.zeros((n, 3), np.float32)
        emit = np.zeros((n, 3), np.float32)
        self._check(self.amd.rt_shade_batch(self.rt_scene, rays.ctypes.data, hits.ctypes.data, n, seed, sc.ctypes.data,
                                            att.ctypes.data, emit.ctypes.data))
        return sc, att, emit

    def primary_rays(self, pixels, sample, seed=0):
        pixels = np.ascontiguousarray(pixels, dtype=np.uint32)
        out = np.zeros(len(pixels), _abi.ray_dtype())
        cam = self.camera_desc()
        self._check(self.amd.rt_primary_rays(self.ctx, C.byref(cam), seed, sample, pixels.ctypes.data, len(pixels), out.ctypes.data))
        return out

    def render_accumulate_host(self, acc, first_sample, n_samples, seed=0):
        """rt_render_accumulate on a host buffer (H x W x 4 float32, updated in place)."""
        cam = self.camera_desc()
        assert acc.dtype == np.float32 and acc.flags.c_contiguous
        self._check(self.amd.rt_render_accumulate(self.rt_scene, C.byref(cam), seed, first_sample, n_samples, acc.ctypes.data))
        return acc

    def render_accumulate_device(self, d_acc, first_sample, n_samples, seed=0):
        cam = self.camera_desc()
        self._check(self.amd.rt_render_accumulate_device(self.rt_scene, C.byref(cam), seed, first_sample, n_samples, d_acc))

    def render_accumulate_multi(self, others, d_acc, first_sample, n_samples, seed=0, split=_abi.SPLIT_SAMPLES):
        """One render call over this scene's GPU and those of `others` (the same scene built on other contexts);
        d_acc lives on this scene's GPU.  Returns after enqueueing; `synchronize()` of this scene waits for all."""
        cam = self.camera_desc()
        scenes = [self] + list(others)
        arr = (C.c_void_p * len(scenes))(*[s.rt_scene for s in scenes])
        self._check(self.amd.rt_render_accumulate_multi(arr, len(scenes), split, C.byref(cam), seed, first_sample, n_samples, d_acc))

    def render_accumulate_device_noalloc(self, d_acc, first_sample, n_samples, seed=0):
        """The variant that allocates nothing: its queues live in the library's fixed device array."""
        cam = self.camera_desc()
        self._check(self.amd.rt_render_accumulate_device_noalloc(self.rt_scene, C.byref(cam), seed, first_sample, n_samples, d_acc))

    def render_accumulate_rows_device(self, d_acc, first_sample, n_samples, first_row, n_rows, seed=0):
        """Rows first_row .. first_row + n_rows - 1 only; d_acc is still the whole-image buffer."""
        cam = self.camera_desc()
        self._check(self.amd.rt_render_accumulate_rows_device(self.rt_scene, C.byref(cam), seed, first_sample, n_samples, first_row, n_rows, d_acc))

    def info(self):
        i = _abi.RtSceneInfo()
        self._check(self.amd.rt_scene_get_info(self.rt_scene, C.byref(i)))
        return {k: getattr(i, k) for k, _ in i._fields_}

    def synchronize(self):
        self._check(self.amd.rt_ctx_synchronize(self.ctx))

    def set_stream(self, hip_stream):
        self._check(self.amd.rt_ctx_set_stream(self.ctx, hip_stream))

    def set_batch_paths(self, n):
        self._check(self.amd.rt_ctx_set_batch_paths(self.ctx, n))

    def set_pipelines(self, n):
        self._check(self.amd.rt_ctx_set_pipelines(self.ctx, n))

    def set_tail_start(self, first_bounce):
        """First bounce the tail kernel takes over (-1: chosen per batch; >= max_depth: never)."""
        self._check(self.amd.rt_ctx_set_tail_start(self.ctx, first_bounce))

    def set_host_buffer_pinning(self, on):
        """Opt in to rt_render_accumulate keeping the caller's buffer registered (see include/rt_abi.h: release it before freeing)."""
        self._check(self.amd.rt_ctx_set_host_buffer_pinning(self.ctx, int(on)))

    def release_host_buffer(self):
        self._check(self.amd.rt_ctx_release_host_buffer(self.ctx))

    def memory_info(self):
        """(free, total) device memory in bytes on this scene's GPU."""
        free, total = C.c_size_t(), C.c_size_t()
        self._check(self.amd.rt_ctx_memory_info(self.ctx, C.byref(free), C.byref(total)))
        return int(free.value), int(total.value)

    def set_counters(self, on):
        self._check(self.amd.rt_ctx_set_counters(self.ctx, int(on)))

    def counters(self, reset=False):
        c = _abi.RtCounters()
        self._check(self.amd.rt_ctx_get_counters(self.ctx, C.byref(c), int(reset)))
        return c.as_dict()

    def set_phase_timing(self, on):
        self._check(self.amd.rt_ctx_set_phase_timing(self.ctx, int(on)))

    def last_render_ms(self):
        ms = C.c_float()
        self._check(self.amd.rt_ctx_last_render_ms(self.ctx, C.byref(ms)))
        return ms.value

    def phase_ms(self):
        """{phase: (milliseconds, launches)} of the last synchronized render."""
        out = {}
        for i, name in enumerate(["raygen", "extend", "shade", "accumulate", "first_leaves", "tail"]):
            ms, n = C.c_float(), C.c_uint32()
            self._check(self.amd.rt_ctx_phase_ms(self.ctx, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, int(n.value))
        return out

    def finalize(self, acc, total_samples, mode=_abi.FINALIZE_SQRT, exposure=1.0):
        acc = np.ascontiguousarray(acc, dtype=np.float32)
        n = acc.size // 4
        if mode == _abi.FINALIZE_TONEMAP_RGBA:
            out = np.empty(acc.shape[:-1], np.uint32)
        else:
            out = np.empty(acc.shape, np.float32)
        self._check(self.amd.rt_finalize(self.ctx, acc.ctypes.data, n, total_samples, mode, exposure, out.ctypes.data))
        return out
