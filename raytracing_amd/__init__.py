"""MI355X-native path-tracing backend for rookieCookies/raytracing-style scenes.

The render loop (ray generation, BVH traversal, shading, accumulation) runs in
hand-written HIP kernels for gfx950 behind the C ABI of ``include/rt_abi.h``;
this package is the thin Python face used by the tests and the benchmark.
"""
from . import _abi  # noqa: F401
from .scene import RenderError, Scene, load_earth, write_exr, write_pfm  # noqa: F401

__all__ = ["Scene", "RenderError", "load_earth", "write_exr", "write_pfm"]
