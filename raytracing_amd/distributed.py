"""Multi-GPU partitioning of a render: sample slicing (or row slicing) + one sum-reduce.

Every (pixel, sample) is independent (the random numbers are a pure function of
them), so a render of S samples on N ranks is: rank r renders the contiguous
sample numbers of `sample_slice(S, r, N)` for ALL pixels into its own W x H x 4
accumulation buffer, then one all-reduce / reduce (sum, f32) over RCCL combines
the buffers.  There is no other exchange step.  The helpers below are pure
functions plus one collective call, so they can be exercised on CPU with gloo.
"""


def sample_slice(total_samples, rank, world_size, first_sample=1):
    """(first_sample_number, count) of this rank's share; shares differ by at most one
    sample and cover [first_sample, first_sample + total_samples) without overlap."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(total_samples, world_size)
    count = base + (1 if rank < extra else 0)
    start = first_sample + rank * base + min(rank, extra)
    return start, count


def row_slice(height, rank, world_size):
    """(first_row, n_rows) of this rank's share of the image rows.  The alternative to sample slicing for calls
    with few samples (interactive frames): every rank renders all samples of its rows into a zeroed whole-image
    buffer (`rt_render_accumulate_rows_device`), and the same sum-reduce assembles the image.  Each pixel is summed
    on one rank only, so the result equals the one-GPU image bit for bit."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(height, world_size)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def reduce_accumulators(acc, dst=None, group=None):
    """Sum the per-rank accumulation buffers (a torch tensor) in place: to every rank
    (dst=None, all-reduce) or to rank `dst` only."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return acc
    if dst is None:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return acc
