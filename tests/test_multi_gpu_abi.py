"""rt_render_accumulate_multi: one render call over several contexts of one process (one per GPU).

A one-GPU box still runs the whole mechanism - shares of samples or rows, private buffers, peer copies, the sum in GPU
order - with several contexts on GPU 0; with two or more GPUs the same test runs across devices as well.
"""
import ctypes as C

import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import _abi

pytestmark = pytest.mark.gpu

W, H, SPP, SEED = 80, 45, 7, 11


def _device_count():
    import torch
    return torch.cuda.device_count()


class DeviceBuffer:
    def __init__(self, scene, shape=(H, W, 4)):
        self.scene, self.shape = scene, shape
        self.bytes = int(np.prod(shape)) * 4
        self.ptr = C.c_void_p()
        scene._check(scene.amd.rt_buffer_create(scene.ctx, self.bytes, C.byref(self.ptr)))
        scene._check(scene.amd.rt_buffer_zero(scene.ctx, self.ptr, self.bytes))

    def get(self):
        out = np.empty(self.shape, np.float32)
        self.scene._check(self.scene.amd.rt_buffer_download(self.scene.ctx, out.ctypes.data, self.ptr, self.bytes))
        return out

    def free(self):
        self.scene._check(self.scene.amd.rt_buffer_destroy(self.scene.ctx, self.ptr))


def _share(total, i, n):
    base, rem = divmod(total, n)
    return i * base + min(i, rem), base + (1 if i < rem else 0)


def _one_gpu(first, count, start=None):
    s = rt.Scene.builtin(8, W, H)
    buf = DeviceBuffer(s)
    if start is not None:
        s._check(s.amd.rt_buffer_upload(s.ctx, buf.ptr, start.ctypes.data, buf.bytes))
    if count:
        s.render_accumulate_device(buf.ptr, first, count, seed=SEED)
    out = buf.get()
    buf.free()
    return out


def _check_multi(devices):
    n = len(devices)
    scenes = [rt.Scene.builtin(8, W, H, device=d) for d in devices]
    head, others = scenes[0], scenes[1:]
    # rows: two samples by GPU 0 alone first, so that the shares are rendered on top of an image that is not zero
    buf = DeviceBuffer(head)
    head.render_accumulate_device(buf.ptr, 1, 2, seed=SEED)
    head.render_accumulate_multi(others, buf.ptr, 3, SPP - 2, seed=SEED, split=_abi.SPLIT_ROWS)
    head.synchronize()
    assert np.array_equal(buf.get(), _one_gpu(1, SPP)), "split by rows must reproduce the one-GPU image bit for bit"
    buf.free()
    # samples: GPU 0's share goes straight into the image, the other shares are added in GPU order
    buf = DeviceBuffer(head)
    head.render_accumulate_device(buf.ptr, 1, 1, seed=SEED)
    head.render_accumulate_multi(others, buf.ptr, 2, SPP - 1, seed=SEED, split=_abi.SPLIT_SAMPLES)
    head.synchronize()
    got = buf.get()
    first, count = _share(SPP - 1, 0, n)
    want = _one_gpu(2 + first, count, start=_one_gpu(1, 1))
    for i in range(1, n):
        first, count = _share(SPP - 1, i, n)
        want[..., :3] += _one_gpu(2 + first, count)[..., :3]
    assert np.array_equal(got, want), "the sum of the shares must be taken in GPU order"
    assert np.allclose(got, _one_gpu(1, SPP), rtol=1e-5, atol=1e-6)
    # and again into the same buffers: nothing of the first call is left in the private buffers
    head._check(head.amd.rt_buffer_zero(head.ctx, buf.ptr, buf.bytes))
    head.render_accumulate_multi(others, buf.ptr, 1, 2, seed=SEED, split=_abi.SPLIT_SAMPLES)
    head.synchronize()
    assert np.allclose(buf.get(), _one_gpu(1, 2), rtol=1e-5, atol=1e-6)
    buf.free()


@pytest.mark.parametrize("n_ctx", [1, 2, 3, 8])   # eight: more shares than the six samples of the second call
def test_multi_entry_point_with_several_contexts_on_one_gpu(n_ctx):
    _check_multi([0] * n_ctx)


def test_multi_entry_point_with_more_shares_than_rows():
    """Five contexts, an image of three rows, one sample: two shares of the rows and four shares of the samples are empty."""
    w, h = 33, 3
    scenes = [rt.Scene.builtin(6, w, h, max_depth=5) for _ in range(5)]
    head, others = scenes[0], scenes[1:]
    alone = rt.Scene.builtin(6, w, h, max_depth=5)
    ref = DeviceBuffer(alone, (h, w, 4))
    alone.render_accumulate_device(ref.ptr, 1, 2, seed=SEED)
    want = ref.get()
    ref.free()
    for split in (_abi.SPLIT_ROWS, _abi.SPLIT_SAMPLES):
        buf = DeviceBuffer(head, (h, w, 4))
        head.render_accumulate_multi(others, buf.ptr, 1, 1, seed=SEED, split=split)
        head.render_accumulate_multi(others, buf.ptr, 2, 1, seed=SEED, split=split)
        head.synchronize()
        assert np.array_equal(buf.get(), want), split     # one sample per call: a single share holds it, nothing is re-associated
        buf.free()


@pytest.mark.parametrize("n_ctx", [2, 3])
def test_back_to_back_sample_split_calls_without_synchronizing(n_ctx):
    """Four sample-split calls in a row, three samples each (every context has a share), nothing waited for in between,
    every context on a stream of its own and the head's stream busy with other work when the calls are made: the other
    contexts are then whole calls ahead of the head, and their next share must not land in the buffer the head has yet
    to add (the analogue of the rows of src/camera.rs:282-304 all finishing before the image is read)."""
    import torch
    streams = [torch.cuda.Stream() for _ in range(n_ctx)]
    scenes = [rt.Scene.builtin(8, W, H) for _ in range(n_ctx)]
    for s, st in zip(scenes, streams):
        s.set_stream(st.cuda_stream)
    head, others = scenes[0], scenes[1:]
    buf = DeviceBuffer(head)
    # work that keeps the head's stream busy: a render of a larger image into a buffer nobody reads
    busy = rt.Scene.builtin(8, 640, 360)
    busy.set_stream(streams[0].cuda_stream)
    busy.render_accumulate_device(busy.device_accumulator, 1, 8, seed=1)
    per_call, calls = 3, 4
    for k in range(calls):
        head.render_accumulate_multi(others, buf.ptr, 1 + per_call * k, per_call, seed=SEED, split=_abi.SPLIT_SAMPLES)
    head.synchronize()
    torch.cuda.synchronize()
    got = buf.get()
    want = np.zeros((H, W, 4), np.float32)
    for k in range(calls):
        first, count = _share(per_call, 0, n_ctx)
        assert count > 0
        want = _one_gpu(1 + per_call * k + first, count, start=want)
        for i in range(1, n_ctx):
            first, count = _share(per_call, i, n_ctx)
            assert count > 0
            want[..., :3] += _one_gpu(1 + per_call * k + first, count)[..., :3]
    assert np.array_equal(got, want), "every call's shares, summed in GPU order, call after call"
    buf.free()
    busy.close()


def test_multi_entry_point_across_gpus():
    n = _device_count()
    if n < 2:
        pytest.skip("needs two GPUs")
    _check_multi(list(range(min(n, 4))))


def test_multi_entry_point_rejects_bad_arguments():
    s = rt.Scene.builtin(6, 16, 16)
    with pytest.raises(rt.RenderError, match="share a context"):
        s.render_accumulate_multi([s], s.device_accumulator, 1, 1)
    cam = s.camera_desc()
    arr = (C.c_void_p * 1)(s.rt_scene)
    assert s.amd.rt_render_accumulate_multi(arr, 0, 0, C.byref(cam), 0, 1, 1, s.device_accumulator) == -1
    assert s.amd.rt_render_accumulate_multi(arr, 1, 7, C.byref(cam), 0, 1, 1, s.device_accumulator) == -1
