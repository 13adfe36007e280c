"""One rank of the one-process-per-GPU render, as a program of its own (started by tests/test_two_ranks_gpu.py before it
makes any GPU call itself):

    python multi_rank_worker.py BACKEND RANK WORLD PORT OUT_DIR DEVICE

The rank renders its `sample_slice` of the job's samples through rt_render_accumulate_device into a buffer on its GPU,
the buffers are summed with one all-reduce (backend nccl = RCCL on the device buffers; backend gloo on host copies of
them - what is left when RCCL will not put two ranks on one GPU), and the rank writes the sum to OUT_DIR.
"""
import datetime
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

SCENE, W, H, SPP, SEED = 8, 80, 45, 7, 11


def main():
    backend, rank, world, port, out_dir, device = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
    import torch
    import torch.distributed as dist

    import raytracing_amd as rt
    from raytracing_amd.distributed import reduce_accumulators, sample_slice

    torch.cuda.set_device(device)
    kw = {"device_id": torch.device("cuda", device)} if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=60), **kw)
    scene = rt.Scene.builtin(SCENE, W, H, device=device)
    stream = torch.cuda.current_stream()
    scene.set_stream(stream.cuda_stream)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    first, count = sample_slice(SPP, rank, world, first_sample=1)
    if count:
        scene.render_accumulate_device(acc.data_ptr(), first, count, seed=SEED)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"share_{rank}.npy"), acc.cpu().numpy())
    if backend == "nccl":
        reduce_accumulators(acc)
        torch.cuda.synchronize()
        total = acc.cpu().numpy()
    else:
        host = acc.cpu()
        reduce_accumulators(host)
        total = host.numpy()
    np.save(os.path.join(out_dir, f"sum_{rank}.npy"), total)
    dist.barrier()
    dist.destroy_process_group()
    scene.close()


if __name__ == "__main__":
    main()
