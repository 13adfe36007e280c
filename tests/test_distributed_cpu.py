"""The N > 1 path on CPU: two gloo ranks slice the samples and sum-reduce their
accumulation buffers.  The renderer inside each rank is the CPU oracle here (no GPU
in this test); the slicing and the collective are the product's."""
import os
import socket
import sys

import numpy as np
import pytest

from raytracing_amd.distributed import row_slice, sample_slice


def test_sample_slices_partition_the_range():
    for total in (1, 2, 7, 8, 4096, 10000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                start, count = sample_slice(total, r, world, first_sample=1)
                seen.extend(range(start, start + count))
            assert seen == list(range(1, total + 1))
            counts = [sample_slice(total, r, world)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        sample_slice(8, 2, 2)


def test_row_slices_partition_the_image():
    for height in (1, 2, 7, 225, 2160):
        for world in (1, 2, 3, 8):
            rows = []
            for r in range(world):
                first, n = row_slice(height, r, world)
                rows.extend(range(first, first + n))
            assert rows == list(range(height))
    with pytest.raises(ValueError):
        row_slice(8, -1, 2)


def _worker(rank, world, port, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import oracle_binding as ob
    from raytracing_amd.distributed import reduce_accumulators, sample_slice
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    scene = ob.OracleScene.builtin(6, 24, 24)
    first, count = sample_slice(7, rank, world)
    acc = scene.render(count, first_sample=first, seed=21, threads=1)
    t = torch.from_numpy(acc)
    reduce_accumulators(t)                # all-reduce: every rank ends up with the sum
    np.save(os.path.join(out_dir, f"all_{rank}.npy"), t.numpy())
    acc2 = torch.from_numpy(scene.render(count, first_sample=first, seed=21, threads=1))
    reduce_accumulators(acc2, dst=0)      # reduce to rank 0 only
    np.save(os.path.join(out_dir, f"root_{rank}.npy"), acc2.numpy())
    # row slicing: every rank contributes all samples of its rows, zeros elsewhere
    from raytracing_amd.distributed import row_slice
    first_row, n_rows = row_slice(24, rank, world)
    full = scene.render(7, first_sample=1, seed=21, threads=1)
    mine = np.zeros_like(full)
    mine[first_row:first_row + n_rows] = full[first_row:first_row + n_rows]
    rows = torch.from_numpy(mine)
    reduce_accumulators(rows)
    np.save(os.path.join(out_dir, f"rows_{rank}.npy"), rows.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sample_slicing_and_reduce(tmp_path):
    import torch.multiprocessing as mp

    import oracle_binding as ob
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    whole = ob.OracleScene.builtin(6, 24, 24).render(7, first_sample=1, seed=21, threads=1)
    a0, a1 = np.load(tmp_path / "all_0.npy"), np.load(tmp_path / "all_1.npy")
    assert np.array_equal(a0, a1)
    # same samples, different summation order: equal up to f32 rounding of the partial sums
    assert np.allclose(a0, whole, rtol=1e-6, atol=1e-6)
    assert np.allclose(np.load(tmp_path / "root_0.npy"), whole, rtol=1e-6, atol=1e-6)
    # row slicing sums each pixel on one rank: identical to the single-process image
    assert np.array_equal(np.load(tmp_path / "rows_0.npy"), whole) and np.array_equal(np.load(tmp_path / "rows_1.npy"), whole)
