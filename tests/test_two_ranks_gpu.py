"""The one-process-per-GPU path with more than one rank on a one-GPU box: two fresh processes share GPU 0, each renders
its slice of the samples through the C ABI, one all-reduce sums the buffers (RCCL if it accepts two ranks on one device,
else gloo on host copies), and the sum is checked against the image one process renders alone.  The reference's
parallel unit is the image row (src/camera.rs:282-304); here it is the sample number."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import raytracing_amd as rt

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "multi_rank_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(backend, world, out_dir):
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, WORKER, backend, str(r), str(world), str(port), str(out_dir), "0"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    ok, logs = True, []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=100)
        except subprocess.TimeoutExpired:
            p.kill()          # the exact process this test started
            out, _ = p.communicate()
            ok = False
        logs.append(out.decode(errors="replace")[-600:])
        ok = ok and p.returncode == 0
    return ok, logs


def test_two_ranks_on_one_gpu_sum_to_the_one_process_image(tmp_path):
    import multi_rank_worker as mw
    from raytracing_amd.distributed import sample_slice
    world = 2
    used = "nccl"
    ok, logs = _run_ranks("nccl", world, tmp_path)
    if not ok:   # RCCL refuses two ranks on one device: the same ranks, the collective over gloo on host copies
        used = "gloo"
        for f in tmp_path.glob("*.npy"):
            f.unlink()
        ok, logs2 = _run_ranks("gloo", world, tmp_path)
        assert ok, "\n".join(logs + logs2)
    print(f"collective backend used: {used}")
    sums = [np.load(tmp_path / f"sum_{r}.npy") for r in range(world)]
    shares = [np.load(tmp_path / f"share_{r}.npy") for r in range(world)]
    assert np.array_equal(sums[0], sums[1]), "all-reduce: every rank holds the same sum"
    # each rank's share is what one process renders for those sample numbers, bit for bit
    alone = rt.Scene.builtin(mw.SCENE, mw.W, mw.H)
    for r in range(world):
        first, count = sample_slice(mw.SPP, r, world, first_sample=1)
        one = rt.Scene.builtin(mw.SCENE, mw.W, mw.H)
        one.set_seed(mw.SEED)
        one.render_accumulate_device(one.device_accumulator, first, count, seed=mw.SEED)
        one.synchronize()
        assert np.array_equal(one.accumulator(), shares[r]), r
    assert np.array_equal(sums[0][..., :3], (shares[0] + shares[1])[..., :3]), "two addends: the sum has one order"
    # and the whole job in one process: the same samples, summed pixel by pixel in sample order instead
    alone.render_accumulate_device(alone.device_accumulator, 1, mw.SPP, seed=mw.SEED)
    alone.synchronize()
    assert np.allclose(sums[0], alone.accumulator(), rtol=1e-5, atol=1e-6)
