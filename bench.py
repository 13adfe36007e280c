#!/usr/bin/env python3
"""Benchmark of the render hot path: Msamples/s on the Book-2 final scene.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch: `--spp-per-step` samples of
every pixel of the 3840x2160 image on each GPU (ray generation, extend, shade,
accumulate for max_depth bounces).  The N ranks render disjoint sample numbers
of the same image (sample slicing); one RCCL all-reduce of the accumulation
buffers closes the timed region.  Inputs (the scene, the queues and the
accumulation buffer) are resident in HBM before the clock starts.

Rank 0 prints one JSON line with the throughput, the roofline position of the
dominant kernel (extend) measured with HIP events on the render stream, and the
CPU baseline (the C++ restatement of the reference, reference random numbers,
host threads) measured on this machine.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, MAX_DEPTH, SCENE_ID = 3840, 2160, 25, 8
JOB_SPP = 4096                       # the configuration BASELINE.json quotes the metric on
HBM_PEAK_GBS = 8000.0                # MI355X HBM3E, spec
FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X FP32 vector, spec
# Algorithmic bytes / flops per unit of work (one path segment = one ray cast + its shading):
# the extend kernel reads one ray record (origin, direction, time, throughput, pixel, sample:
# 48 B) and writes one hit record (t, primitive, u, v: 16 B).
EXTEND_BYTES_PER_SEGMENT = 64
# a node visit slab-tests the four child boxes of a node: 4 x 25 flops (the count for two-child nodes was 2 x 25)
FLOPS = {"node_visits": 100, "sphere_tests": 30, "quad_tests": 45, "hits": 20}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=6)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--spp-per-step", type=int, default=32)   # two batches of 16 spp (133 M paths each), one per pipeline
    p.add_argument("--width", type=int, default=WIDTH)
    p.add_argument("--height", type=int, default=HEIGHT)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-spp", type=int, default=384, help="samples per pixel of the 480x270 CPU baseline render")
    return p.parse_args()


def cpu_baseline(spp):
    """The oracle in reference-faithful mode (per-row xoshiro256++, median-split BVH2,
    explicit-stack traversal) on all host threads this job may use."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    w, h = 480, 270
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    scene = ob.OracleScene.builtin(SCENE_ID, w, h, max_depth=MAX_DEPTH)
    scene.render(1, first_sample=1, mode=ob.RNG_XOSHIRO, threads=threads)  # touch everything once
    t0 = time.perf_counter()
    scene.render(spp, first_sample=2, mode=ob.RNG_XOSHIRO, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": round(w * h * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"Book-2 final scene {w}x{h}, {spp} spp, depth {MAX_DEPTH} ({w * h * spp / 1e6:.2f} Msamples in {dt:.1f} s); "
                      "C++ restatement of the reference with its per-row xoshiro256++ stream, not the Rust binary"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 needs one process per GPU: launch with "
                             "python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import raytracing_amd as rt
    from raytracing_amd.distributed import reduce_accumulators, sample_slice

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    w, h, spp = args.width, args.height, args.spp_per_step
    scene = rt.Scene.builtin(SCENE_ID, w, h, max_depth=MAX_DEPTH, device=local_rank)
    stream = torch.cuda.current_stream()
    scene.set_stream(stream.cuda_stream)
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")

    # sample numbers of this rank: a contiguous slice of the whole job
    per_rank = (args.warmup + args.steps) * spp
    first, count = sample_slice(per_rank * world, rank, world, first_sample=1)
    assert count == per_rank
    nxt = first

    def step():
        nonlocal nxt
        scene.render_accumulate_device(acc.data_ptr(), nxt, spp, seed=args.seed)
        nxt += spp

    for _ in range(args.warmup):
        step()
    reduce_accumulators(acc)           # untimed: the first collective of this size sets up the communicator's channels
    torch.cuda.synchronize()
    acc.zero_()
    scene.set_phase_timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    reduce_accumulators(acc)           # the job's one exchange step: sum of the per-rank buffers
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    scene.synchronize()
    phases = scene.phase_ms()          # HIP events around every launch of the timed region (this rank's stream)
    scene.set_phase_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples = float(w) * h * spp * args.steps * world
    value = samples / elapsed / 1e6

    result = None
    if rank == 0:
        # work counters of one step's sample numbers (untimed counting build of the extend kernel)
        scene.set_counters(True)
        scratch = torch.zeros_like(acc)
        scene.render_accumulate_device(scratch.data_ptr(), nxt - spp, spp, seed=args.seed)
        torch.cuda.synchronize()
        cnt = scene.counters(reset=True)
        scene.set_counters(False)
        del scratch
        per_segment = {k: round(cnt[k] / max(1, cnt["segments"]), 3) for k in ("node_visits", "sphere_tests", "quad_tests", "medium_tests", "hits")}
        # the same step once more with one pipeline, so that no other kernel shares the chip with k_extend
        scene.set_pipelines(1)
        scene.set_phase_timing(True)
        iso = torch.zeros_like(acc)
        scene.render_accumulate_device(iso.data_ptr(), nxt - spp, spp, seed=args.seed)
        torch.cuda.synchronize()
        scene.synchronize()
        iso_ms, iso_launches = scene.phase_ms()["extend"]
        scene.set_phase_timing(False)
        scene.set_pipelines(2)
        del iso
        ext_ms, ext_launches = phases["extend"]
        # segments of the whole timed region: the counted step times the number of steps (steps differ
        # only in their sample numbers; their segment counts agree to ~0.1 %)
        segments = cnt["segments"] * args.steps
        for k in FLOPS:
            cnt[k] *= args.steps
        cnt["samples"] *= args.steps
        ext_bytes = EXTEND_BYTES_PER_SEGMENT * segments
        ext_flops = sum(FLOPS[k] * cnt[k] for k in FLOPS)
        achieved_gbs = ext_bytes / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
        achieved_tf = ext_flops / (ext_ms * 1e-3) / 1e12 if ext_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "extend_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"kernel": "k_extend", "bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "bytes_per_launch": round(ext_bytes / max(1, ext_launches), 1), "launches": ext_launches,
                    "avg_launch_ms": round(ext_ms / max(1, ext_launches), 5), "segments_per_sample": round(segments / cnt["samples"], 4),
                    "alone": {"note": "one step with a single pipeline: k_extend does not share the chip with the other batch's kernels",
                              "avg_launch_ms": round(iso_ms / max(1, iso_launches), 5),
                              "achieved": round(EXTEND_BYTES_PER_SEGMENT * cnt["segments"] / (iso_ms * 1e-3) / 1e9, 2) if iso_ms > 0 else 0.0},
                    "valu": {"achieved": round(achieved_tf, 3), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(achieved_tf / FP32_VECTOR_PEAK_TFLOPS, 5)},
                    "work_per_segment": per_segment, "phase_ms": {k: round(v[0], 3) for k, v in phases.items()}}
        result = {
            "metric": "Msamples/sec (whole node) on Book-2 final scene", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Book-2 final scene (reference scene 8, 3410 leaves) {w}x{h}, max depth {MAX_DEPTH}, "
                                   f"{spp} spp per step per GPU, sample-sliced over {world} GPU(s) of the {JOB_SPP}-spp job, "
                                   "one all-reduce of the accumulation buffer",
                       "width": w, "height": h, "spp_per_step": spp, "max_depth": MAX_DEPTH, "parallelism": f"sample-slice x{world}"},
            "roofline": roofline,
            "scene": scene.info(),
        }
        if not args.no_cpu_baseline and world == 1:  # the host cores are timed on single-GPU runs only
            result["cpu_baseline"] = cpu_baseline(args.cpu_spp)
            result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
