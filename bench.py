#!/usr/bin/env python3
"""Benchmark of the render hot path: Msamples/s on the Book-2 final scene.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch: `--spp-per-step` samples of
every pixel of the 3840x2160 image on each GPU (ray generation, extend, shade,
accumulate for max_depth bounces).  The N ranks render disjoint sample numbers
of the same image (sample slicing); one RCCL all-reduce of the accumulation
buffers closes the timed region.  Inputs (the scene, the queues and the
accumulation buffer) are resident in HBM before the clock starts, and nothing
but the render calls and the collective runs inside it.

After the timed region rank 0 repeats one step three times, untimed: with HIP
events around every launch (per-kernel times: the roofline position of the
dominant kernel, extend), with the counting build of the kernels (work per
segment), and with a single pipeline (extend alone on the chip).  On a
single-GPU run it then measures the benchmark's other configurations and
the CPU baseline (the C++ restatement of the reference with the reference's
random numbers, on this machine's host cores) and prints ONE JSON line.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, MAX_DEPTH, SCENE_ID = 3840, 2160, 25, 8
JOB_SPP = 4096                       # the configuration the metric is quoted on
HBM_PEAK_GBS = 8000.0                # MI355X HBM3E
FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X FP32 vector
# Algorithmic bytes per unit of work (one path segment = one ray cast + its shading), the counting this
# benchmark's roofline figures use throughout:
#   extend kernel      reads one ray record (origin, direction, time, throughput, pixel, sample: 48 B),
#                      writes one hit record (t, primitive, u, v: 16 B)                           -> 64 B / segment
#   whole pipeline     ray record written + read (2 x 48 B), hit record written + read (2 x 16 B)  -> 128 B / segment
#                      plus the accumulation buffer read + written once per sample                 -> 32 B / sample
#   scene gathers      (what dominates when the tree does not fit the caches: 10 M spheres)
#                      one 128-byte node of four children per node visit, 16 B per sphere test
EXTEND_BYTES_PER_SEGMENT = 64
PIPELINE_BYTES_PER_SEGMENT, PIPELINE_BYTES_PER_SAMPLE = 128, 32
NODE_BYTES, SPHERE_BYTES = 128, 16
# a node visit slab-tests the four child boxes of a node: 4 x 25 flops
FLOPS = {"node_visits": 100, "sphere_tests": 30, "quad_tests": 45, "hits": 20}

# The benchmark's other configurations on one GPU: (scene id, width, height, depth, samples per call, extra)
CONFIGS = {
    "C1 Book-1 random spheres 400x225, depth 50, 100 spp per call": (0, 400, 225, 50, 100, {}),
    "C2 earth + perlin spheres 1920x1080, depth 25, 50 spp per call": (100, 1920, 1080, 25, 50, {}),
    "C3 Cornell box 1024x1024, depth 25, 64 spp per call": (6, 1024, 1024, 25, 64, {}),
    "C5 10 M random spheres 1920x1080, primary + 1 bounce, 16 spp per call": (102, 1920, 1080, 2, 16, {"stress_count": 10_000_000}),
}


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
    p.add_argument("--no-configs", action="store_true", help="skip the benchmark's other configurations")
    p.add_argument("--cpu-spp", type=int, default=384, help="samples per pixel of the 480x270 CPU baseline render")
    return p.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def native_oracle():
    """The oracle built for this machine's host cores (-O3 -march=native), in a temporary directory; None when there
    is no compiler here (the portable build that ships with the repository is used then)."""
    cxx = shutil.which("g++")
    if not cxx:
        return None, "portable build (-march=x86-64-v3): no g++ on this machine"
    out = os.path.join(tempfile.mkdtemp(prefix="rt_oracle_"), "liboracle_native.so")
    cmd = [cxx, "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-shared", "-o", out,
           os.path.join(ROOT, "oracle", "oracle_capi.cpp")]
    try:
        subprocess.run(cmd, check=True, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (subprocess.SubprocessError, OSError):
        return None, "portable build (-march=x86-64-v3): the -march=native build failed here"
    return out, "-O3 -march=native build made on this machine"


def cpu_baseline(spp):
    """The oracle in reference-faithful mode (per-row xoshiro256++, median-split BVH2, explicit-stack traversal) on
    all host threads this job may use."""
    lib, how = native_oracle()
    if lib:
        os.environ["RT_ORACLE_LIBRARY"] = lib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    w, h = 480, 270
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    scene = ob.OracleScene.builtin(SCENE_ID, w, h, max_depth=MAX_DEPTH)
    scene.render(1, first_sample=1, mode=ob.RNG_XOSHIRO, threads=threads)  # touch everything once
    t0 = time.perf_counter()
    scene.render(spp, first_sample=2, mode=ob.RNG_XOSHIRO, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": round(w * h * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "build": how,
            "sample": f"Book-2 final scene {w}x{h}, {spp} spp, depth {MAX_DEPTH} ({w * h * spp / 1e6:.2f} Msamples in {dt:.1f} s); "
                      "C++ restatement of the reference with its per-row xoshiro256++ stream, not the Rust binary"}


def other_configs(rt, device, traffic):
    """Throughput of the benchmark's other configurations: wall clock of whole render calls on one GPU."""
    out = {}
    for name, (sid, w, h, depth, spp, extra) in CONFIGS.items():
        t0 = time.perf_counter()
        scene = rt.Scene.builtin(sid, w, h, max_depth=depth, device=device, **extra)
        scene.render_samples(1)
        scene.synchronize()
        t_setup = time.perf_counter() - t0
        scene.set_counters(True)
        scene.render_samples(1)
        scene.synchronize()
        cnt = scene.counters(reset=True)
        scene.set_counters(False)
        scene.render_samples(spp)
        scene.render_samples(spp)
        scene.synchronize()
        calls = 3
        t0 = time.perf_counter()
        for _ in range(calls):
            scene.render_samples(spp)
        scene.synchronize()
        dt = (time.perf_counter() - t0) / calls
        seg = cnt["segments"] / max(1, cnt["samples"])
        info = scene.info()
        row = {"msamples_per_s": round(w * h * spp / dt / 1e6, 1), "mrays_per_s": round(w * h * spp * seg / dt / 1e6, 1),
               "ms_per_call": round(dt * 1e3, 3), "segments_per_sample": round(seg, 3),
               "node_visits_per_segment": round(cnt["node_visits"] / max(1, cnt["segments"]), 2),
               "leaves": info["leaves_in_tree"] + info["leaves_tested_first"], "bvh_nodes": info["bvh_nodes"],
               "scene_bytes": info["device_bytes"], "setup_s": round(t_setup, 2)}
        if sid == 102:
            # scene gathers: the tree (640 MB of nodes, 160 MB of spheres) does not fit the caches, every node visit and
            # sphere test is a gather; algorithmic bytes against the HBM peak, and what the counters saw of them
            per_seg = NODE_BYTES * cnt["node_visits"] / max(1, cnt["segments"]) + SPHERE_BYTES * cnt["sphere_tests"] / max(1, cnt["segments"])
            gbs = per_seg * w * h * spp * seg / dt / 1e9
            row["roofline"] = {"bound": "hbm", "bytes": "128 B x node visits + 16 B x sphere tests (gathers)", "bytes_per_segment": round(per_seg, 1),
                               "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                               "measured": (traffic or {}).get("c5")}
        out[name] = row
        scene.close()
    return out


def main():
    args = parse_args()
    # The one JSON line is all this program writes to its standard output: whatever libraries print there (RCCL writes a
    # version banner when its first communicator comes up) goes to standard error instead.
    sys.stdout.flush()
    json_out = os.dup(1)
    os.dup2(2, 1)
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
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run, also with one rank
    if world > 1 or launched:
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
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    reduce_accumulators(acc)           # the job's one exchange step: sum of the per-rank buffers
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples = float(w) * h * spp * args.steps * world
    value = samples / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    result = None
    if rank == 0:
        scratch = torch.zeros_like(acc)
        # one more step, untimed, with a HIP event pair around every launch (on the pipelines' own streams): per-kernel times
        # of exactly what the timed region ran
        scene.set_phase_timing(True)
        scene.render_accumulate_device(scratch.data_ptr(), nxt - spp, spp, seed=args.seed)
        torch.cuda.synchronize()
        scene.synchronize()
        phases = scene.phase_ms()
        scene.set_phase_timing(False)
        # the same step with the counting build of the kernels: work per segment
        scene.set_counters(True)
        scene.render_accumulate_device(scratch.data_ptr(), nxt - spp, spp, seed=args.seed)
        torch.cuda.synchronize()
        cnt = scene.counters(reset=True)
        scene.set_counters(False)
        per_segment = {k: round(cnt[k] / max(1, cnt["segments"]), 3) for k in ("node_visits", "sphere_tests", "quad_tests", "medium_tests", "hits")}
        # and with one pipeline, so that no other kernel shares the chip with k_extend
        scene.set_pipelines(1)
        scene.set_phase_timing(True)
        scene.render_accumulate_device(scratch.data_ptr(), nxt - spp, spp, seed=args.seed)
        torch.cuda.synchronize()
        scene.synchronize()
        iso_ms, iso_launches = scene.phase_ms()["extend"]
        scene.set_phase_timing(False)
        scene.set_pipelines(2)
        del scratch
        ext_ms, ext_launches = phases["extend"]
        segments = cnt["segments"]                       # of one step
        ext_bytes = EXTEND_BYTES_PER_SEGMENT * segments
        ext_flops = sum(FLOPS[k] * cnt[k] for k in FLOPS)
        achieved_gbs = ext_bytes / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
        achieved_tf = ext_flops / (ext_ms * 1e-3) / 1e12 if ext_ms > 0 else 0.0
        traffic = {}
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile))
            except Exception:
                traffic = {}
        pipe_bytes = PIPELINE_BYTES_PER_SEGMENT * segments + PIPELINE_BYTES_PER_SAMPLE * cnt["samples"]
        pipe_gbs = pipe_bytes / (ms_per_step * 1e-3) / 1e9
        measured_per_segment = (traffic.get("pipeline") or {}).get("hbm_bytes_per_segment")
        roofline = {"kernel": "k_extend", "bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": (traffic.get("extend") or {}).get("hbm_bytes_per_launch"),
                    "bytes_per_launch": round(ext_bytes / max(1, ext_launches), 1), "launches": ext_launches,
                    "avg_launch_ms": round(ext_ms / max(1, ext_launches), 5), "segments_per_sample": round(segments / cnt["samples"], 4),
                    "measured": "HIP events around every launch of one more step after the timed region (same sample numbers per step, two pipelines)",
                    "alone": {"note": "one step with a single pipeline: k_extend does not share the chip with the other batch's kernels",
                              "avg_launch_ms": round(iso_ms / max(1, iso_launches), 5),
                              "achieved": round(ext_bytes / (iso_ms * 1e-3) / 1e9, 2) if iso_ms > 0 else 0.0,
                              "frac": round(ext_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if iso_ms > 0 else 0.0},
                    "valu": {"achieved": round(achieved_tf, 3), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(achieved_tf / FP32_VECTOR_PEAK_TFLOPS, 5),
                             "issue": traffic.get("extend_issue")},  # what the kernel is bound by: instruction issue x lane occupancy (PMC)
                    "pipeline": {"note": "the whole step against the HBM peak: (128 B x segments + 32 B x samples) / ms_per_step of the timed region",
                                 "bytes_per_step": pipe_bytes, "achieved": round(pipe_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(pipe_gbs / HBM_PEAK_GBS, 4),
                                 "algorithmic_bytes_per_segment": round(pipe_bytes / segments, 1),
                                 "measured_hbm_bytes_per_segment": measured_per_segment,
                                 "traffic_ratio": round(measured_per_segment / (pipe_bytes / segments), 3) if measured_per_segment else None},
                    "work_per_segment": per_segment, "phase_ms": {k: round(v[0], 3) for k, v in phases.items()}}
        result = {
            "metric": "Msamples/sec (whole node) on Book-2 final scene", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Book-2 final scene (reference scene 8, 3410 leaves) {w}x{h}, max depth {MAX_DEPTH}, "
                                   f"{spp} spp per step per GPU, sample-sliced over {world} GPU(s) of the {JOB_SPP}-spp job, "
                                   "one all-reduce of the accumulation buffer",
                       "width": w, "height": h, "spp_per_step": spp, "max_depth": MAX_DEPTH, "parallelism": f"sample-slice x{world}"},
            "roofline": roofline,
            "scene": scene.info(),
        }
        if world == 1 and not args.no_configs:
            scene.close()
            del acc
            torch.cuda.empty_cache()
            result["configs"] = other_configs(rt, local_rank, traffic)
        if not args.no_cpu_baseline and world == 1:  # the host cores are timed on single-GPU runs only
            result["cpu_baseline"] = cpu_baseline(args.cpu_spp)
            result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_out, (json.dumps(result) + "\n").encode())


if __name__ == "__main__":
    main()
