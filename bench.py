#!/usr/bin/env python3
"""Benchmark of the render hot path: Msamples/s on the Book-2 final scene.

    python bench.py --gpus N --steps K --warmup W [--config c4|c3] [--scaling weak|strong]

One step = one pass of the hot path over one batch: `--spp-per-step` samples of
every pixel of the 3840x2160 image on each GPU (ray generation, extend, shade,
accumulate for max_depth bounces).  The N ranks render disjoint sample numbers
of the same image (sample slicing); one RCCL all-reduce of the accumulation
buffers closes the timed region.  Inputs (the scene, the queues and the
accumulation buffer) are resident in HBM before the clock starts, and nothing
but the render calls and the collective runs inside it.

`--scaling strong` fixes the whole job instead of the work per GPU: the 4096
samples per pixel of the headline job are sliced over the N ranks, each rank
renders its slice in K steps.  `--config c3` is the Cornell box at 1024x1024,
10 000 samples per pixel, always sliced that way (8 x 1250 on eight GPUs), with
its all-reduce of 16.8 MB timed separately.

Everything in the printed line is measured in the run that prints it, with one
exception that says so: the object `from_profiles` repeats figures of committed
rocprofv3 runs (file and commit named there).  After the timed region rank 0
repeats one step three times, untimed: with HIP events around every launch
(per-kernel times: the roofline position of the dominant kernel, extend), with
the counting build of the kernels (work per segment), and with a single
pipeline (extend alone on the chip).  On a single-GPU run it then collects the
extend kernel's memory-side traffic with two rocprofv3 counter passes over a
child process that renders the same batch, measures the benchmark's other
configurations and the CPU baseline (the C++ restatement of the reference with
the reference's random numbers, on this machine's host cores) and prints ONE
JSON line.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                # MI355X HBM3E
FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X FP32 vector
# The configurations this program times as the whole-job metric: scene id, image, depth, samples per pixel of the job
JOBS = {
    "c4": {"scene": 8, "width": 3840, "height": 2160, "max_depth": 25, "job_spp": 4096,
           "name": "Book-2 final scene (reference scene 8, 3410 leaves)"},
    "c3": {"scene": 6, "width": 1024, "height": 1024, "max_depth": 25, "job_spp": 10000,
           "name": "Cornell box (reference scene 6, 18 quads)"},
}
# Algorithmic bytes per unit of work (one path segment = one ray cast + its shading), the counting this
# benchmark's roofline figures use throughout:
#   extend kernel      reads one ray record (origin, direction, time, throughput, pixel, sample: 48 B),
#                      writes one hit record (t, primitive, u, v: 16 B)                           -> 64 B / segment
#   whole pipeline     ray record written + read (2 x 48 B), hit record written + read (2 x 16 B)  -> 128 B / segment
#                      plus the accumulation buffer read + written once per sample                 -> 32 B / sample
EXTEND_BYTES_PER_SEGMENT = 64
PIPELINE_BYTES_PER_SEGMENT, PIPELINE_BYTES_PER_SAMPLE = 128, 32
# a node visit slab-tests the four child boxes of a node: 4 x 25 flops
FLOPS = {"node_visits": 100, "sphere_tests": 30, "quad_tests": 45, "hits": 20}

# The benchmark's other configurations on one GPU: (scene id, width, height, depth, samples per call, extra)
CONFIGS = {
    "C1 Book-1 random spheres 400x225, depth 50, 100 spp per call": (0, 400, 225, 50, 100, {}),
    "C2 earth + perlin spheres 1920x1080, depth 25, 50 spp per call": (100, 1920, 1080, 25, 50, {}),
    "C3 Cornell box 1024x1024, depth 25, 64 spp per call": (6, 1024, 1024, 25, 64, {}),
    "C5 10 M random spheres 1920x1080, primary + 1 bounce, 16 spp per call": (102, 1920, 1080, 2, 16, {"stress_count": 10_000_000}),
}
PMC_SPP = 16   # samples per pixel of the batch the counter passes render: the batch of one pipeline in the timed region


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=6)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--config", choices=sorted(JOBS), default="c4")
    p.add_argument("--scaling", choices=["weak", "strong"], default=None, help="default: weak for c4, strong for c3")
    p.add_argument("--spp-per-step", type=int, default=32)   # weak scaling: two batches of 16 spp (133 M paths each), one per pipeline
    p.add_argument("--job-spp", type=int, default=None, help="strong scaling: samples per pixel of the whole job (default: the configuration's)")
    p.add_argument("--width", type=int, default=None)
    p.add_argument("--height", type=int, default=None)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-configs", action="store_true", help="skip the benchmark's other configurations")
    p.add_argument("--no-live-traffic", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic is null then)")
    p.add_argument("--cpu-spp", type=int, default=384, help="samples per pixel of the 480x270 CPU baseline render")
    p.add_argument("--pmc-workload", action="store_true", help=argparse.SUPPRESS)   # the child process of the counter passes
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------
# host CPU: what it is and how much of it this process may use
# ---------------------------------------------------------------------------------------------------------
def cpu_info():
    model, logical = "unknown", 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                logical += 1
    except OSError:
        pass
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None   # cgroup v2 CPU bandwidth limit, in CPUs
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    share = None   # what the job scheduler says this job's share of the machine is
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            share = int(v)
            break
    return {"model_name": model, "logical_cpus_in_cpuinfo": logical, "os_cpu_count": os.cpu_count(), "affinity_count": affinity,
            "cgroup_cpu_quota": quota, "job_cpu_share_env": share}


def cpu_threads_to_use(info):
    """All host threads this process may use: its affinity mask, cut down to the cgroup's bandwidth limit or the job's CPU
    share where one is set (on a shared host, threads beyond that share are another job's cores)."""
    n = info["affinity_count"]
    if info["cgroup_cpu_quota"]:
        n = min(n, max(1, int(info["cgroup_cpu_quota"])))
    if info["job_cpu_share_env"]:
        n = min(n, info["job_cpu_share_env"])
    return max(1, n)


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
    info = cpu_info()
    threads = cpu_threads_to_use(info)
    job = JOBS["c4"]
    scene = ob.OracleScene.builtin(job["scene"], w, h, max_depth=job["max_depth"])
    scene.render(1, first_sample=1, mode=ob.RNG_XOSHIRO, threads=threads)  # touch everything once
    t0 = time.perf_counter()
    scene.render(spp, first_sample=2, mode=ob.RNG_XOSHIRO, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": round(w * h * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port", "cpu": info["model_name"],
            "host": info, "build": how,
            "sample": f"Book-2 final scene {w}x{h}, {spp} spp, depth {job['max_depth']} ({w * h * spp / 1e6:.2f} Msamples in {dt:.1f} s) on "
                      f"{threads} threads (affinity mask {info['affinity_count']} of {info['os_cpu_count']} CPUs, cgroup quota "
                      f"{info['cgroup_cpu_quota']}, job share {info['job_cpu_share_env']}); "
                      "C++ restatement of the reference with its per-row xoshiro256++ stream, not the Rust binary"}


# ---------------------------------------------------------------------------------------------------------
# memory-side traffic of the extend kernel, measured by rocprofv3 counter passes over a child process
# ---------------------------------------------------------------------------------------------------------
def pmc_workload(args):
    """What the child process of a counter pass runs (no torch): one batch of the headline scene - PMC_SPP samples of every
    pixel, per-bounce kernels on one pipeline - exactly the batch each pipeline of the timed region renders per step."""
    import raytracing_amd as rt
    job = JOBS["c4"]
    scene = rt.Scene.builtin(job["scene"], args.width or job["width"], args.height or job["height"], max_depth=job["max_depth"])
    scene.set_pipelines(1)
    scene.render_samples(PMC_SPP)
    scene.synchronize()


def live_extend_traffic(args):
    """FETCH_SIZE and WRITE_SIZE of every k_extend launch of one PMC_SPP-sample batch, one rocprofv3 run per counter (they do
    not fit one pass).  Memory-side bytes per launch, corrected as gfx950 needs: FETCH_SIZE counts the 128-byte requests of
    16-byte-per-lane loads as 64 bytes, so it is doubled; WRITE_SIZE is exact; both are in KiB."""
    prof = shutil.which("rocprofv3")
    if not prof:
        return None, {"error": "rocprofv3 not found on this machine"}
    out = tempfile.mkdtemp(prefix="rt_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    sums, launches, seconds = {}, 0, 0.0
    t_all = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, counter)
        cmd = ["timeout", "-s", "KILL", "150", prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.abspath(__file__), "--pmc-workload"]
        if args.width:
            cmd += ["--width", str(args.width)]
        if args.height:
            cmd += ["--height", str(args.height)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=170)
        except (subprocess.SubprocessError, OSError) as e:
            return None, {"error": f"{counter} pass did not finish: {e}"}
        if r.returncode != 0:
            return None, {"error": f"{counter} pass exited with {r.returncode}", "tail": r.stdout.decode(errors="replace")[-300:]}
        total, n = 0.0, 0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_extend" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    total += float(row["Counter_Value"])
                    n += 1
        if n == 0:
            return None, {"error": f"{counter} pass wrote no k_extend rows"}
        sums[counter], launches = total, n
        if counter == "FETCH_SIZE":
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_extend" in row["Kernel_Name"]:
                        seconds += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    shutil.rmtree(out, ignore_errors=True)
    fetch, write = sums["FETCH_SIZE"] * 1024 * 2, sums["WRITE_SIZE"] * 1024
    per_launch = (fetch + write) / launches
    detail = {"how": f"two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a child process of this run that renders one {PMC_SPP}-spp batch "
                     "on one pipeline; FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024, summed over the k_extend launches, / launches",
              "launches": launches, "fetch_bytes": fetch, "write_bytes": write,
              "avg_launch_ms_under_counters": round(seconds / launches * 1e3, 4) if seconds else None,
              "seconds_spent": round(time.perf_counter() - t_all, 1)}
    return per_launch, detail


def from_profiles():
    """Figures of committed rocprofv3 runs (profiles/traffic.json), repeated for convenience: NOT measured in this run."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(tfile))
    except Exception:
        return None
    return {"note": "constants copied from a committed profile of an earlier run of the same kernels on another box; nothing in this object "
                    "was measured by the run that printed this line",
            "file": "profiles/traffic.json", "taken_at_commit": t.get("commit"), "taken_on": t.get("date"),
            "extend": t.get("extend"), "extend_issue": t.get("extend_issue"), "pipeline": t.get("pipeline"), "c5": t.get("c5")}


def other_configs(rt, device):
    """Throughput of the benchmark's other configurations: wall clock of whole render calls on one GPU, and the extend
    kernel's own time (HIP events around its launches in one more call)."""
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
        scene.set_phase_timing(True)
        scene.render_samples(spp)
        scene.synchronize()
        ext_ms, ext_launches = scene.phase_ms()["extend"]
        scene.set_phase_timing(False)
        seg = cnt["segments"] / max(1, cnt["samples"])
        info = scene.info()
        row = {"msamples_per_s": round(w * h * spp / dt / 1e6, 1), "mrays_per_s": round(w * h * spp * seg / dt / 1e6, 1),
               "ms_per_call": round(dt * 1e3, 3), "segments_per_sample": round(seg, 3),
               "node_visits_per_segment": round(cnt["node_visits"] / max(1, cnt["segments"]), 2),
               "extend_ms_per_call": round(ext_ms, 3), "extend_launches_per_call": ext_launches,
               "leaves": info["leaves_in_tree"] + info["leaves_tested_first"], "bvh_nodes": info["bvh_nodes"],
               "scene_bytes": info["device_bytes"], "setup_s": round(t_setup, 2)}
        if sid == 102 and ext_ms > 0:
            # the tree (640 MB of nodes, 160 MB of spheres) does not fit the caches: the kernel's pace in node visits
            # (no roofline fraction is claimed here: a count of gathered bytes includes what the caches absorb, and the
            # memory-side traffic needs counters - see from_profiles.c5)
            visits = cnt["node_visits"] / max(1, cnt["samples"]) * w * h * spp
            row["extend_node_visits_per_s"] = round(visits / (ext_ms * 1e-3) / 1e9, 2)
            row["extend_node_visits_unit"] = "G node visits / s (64-byte nodes), k_extend's own time"
        out[name] = row
        scene.close()
    return out


def main():
    args = parse_args()
    if args.pmc_workload:
        pmc_workload(args)
        return
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

    job = JOBS[args.config]
    scaling = args.scaling or ("strong" if args.config == "c3" else "weak")
    w, h, max_depth = args.width or job["width"], args.height or job["height"], job["max_depth"]
    scene = rt.Scene.builtin(job["scene"], w, h, max_depth=max_depth, device=local_rank)
    stream = torch.cuda.current_stream()
    scene.set_stream(stream.cuda_stream)
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")

    # Sample numbers of this rank: a contiguous slice of the whole job.
    #   weak:   every rank renders spp_per_step samples per step; the job grows with the number of ranks
    #   strong: the job's samples are sliced over the ranks; a rank renders its slice in `steps` steps
    if scaling == "weak":
        spp = args.spp_per_step
        step_spp = [spp] * args.steps
        first, count = sample_slice(spp * args.steps * world, rank, world, first_sample=1)
        assert count == spp * args.steps
        job_samples_per_pixel = spp * args.steps * world
    else:
        job_spp = args.job_spp or job["job_spp"]
        first, count = sample_slice(job_spp, rank, world, first_sample=1)
        base, extra = divmod(count, args.steps)
        step_spp = [base + (1 if i < extra else 0) for i in range(args.steps)]
        spp = max(step_spp) if step_spp else 0
        job_samples_per_pixel = job_spp
    warm_spp = max(1, spp)

    # warm-up: W steps of the same size into the same buffer (sample numbers beyond the job's), then the buffer is zeroed
    warm_first = 1 + job_samples_per_pixel + rank * args.warmup * warm_spp
    for i in range(args.warmup):
        scene.render_accumulate_device(acc.data_ptr(), warm_first + i * warm_spp, warm_spp, seed=args.seed)
    reduce_accumulators(acc)           # untimed: the first collective of this size sets up the communicator's channels
    torch.cuda.synchronize()
    acc.zero_()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nxt = first
    for n in step_spp:
        if n:
            scene.render_accumulate_device(acc.data_ptr(), nxt, n, seed=args.seed)
        nxt += n
    torch.cuda.synchronize()
    t_rendered = time.perf_counter()   # this rank's share is in its buffer
    reduce_accumulators(acc)           # the job's one exchange step: sum of the per-rank buffers
    torch.cuda.synchronize()
    t_reduced = time.perf_counter()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    render_ms, reduce_ms = (t_rendered - t0) * 1e3, (t_reduced - t_rendered) * 1e3
    per_rank = [[render_ms, reduce_ms]]
    allreduce_alone_ms = None
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([render_ms, reduce_ms], dtype=torch.float64, device="cuda")
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        per_rank = [[float(x) for x in e.tolist()] for e in everyone]
        # the collective alone: every rank enters it at the same time (the figure above includes waiting for the slowest rank)
        scratch = torch.zeros_like(acc)
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reduce_accumulators(scratch)
        torch.cuda.synchronize()
        allreduce_alone_ms = (time.perf_counter() - t1) * 1e3
        del scratch

    samples = float(w) * h * job_samples_per_pixel
    value = samples / elapsed / 1e6
    ms_per_step = elapsed / max(1, args.steps) * 1e3

    result = None
    if rank == 0:
        scratch = torch.zeros_like(acc)
        one_step = (first, max(1, spp))
        # one more step, untimed, with a HIP event pair around every launch (on the pipelines' own streams): per-kernel times
        # of exactly what the timed region ran
        scene.set_phase_timing(True)
        scene.render_accumulate_device(scratch.data_ptr(), *one_step, seed=args.seed)
        torch.cuda.synchronize()
        scene.synchronize()
        phases = scene.phase_ms()
        scene.set_phase_timing(False)
        # the same step with the counting build of the kernels: work per segment
        scene.set_counters(True)
        scene.render_accumulate_device(scratch.data_ptr(), *one_step, seed=args.seed)
        torch.cuda.synchronize()
        cnt = scene.counters(reset=True)
        scene.set_counters(False)
        per_segment = {k: round(cnt[k] / max(1, cnt["segments"]), 3) for k in ("node_visits", "sphere_tests", "quad_tests", "medium_tests", "hits")}
        # and with one pipeline, so that no other kernel shares the chip with k_extend
        scene.set_pipelines(1)
        scene.set_phase_timing(True)
        scene.render_accumulate_device(scratch.data_ptr(), *one_step, seed=args.seed)
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
        pipe_bytes = PIPELINE_BYTES_PER_SEGMENT * segments + PIPELINE_BYTES_PER_SAMPLE * cnt["samples"]
        step_ms_of_one_step = ms_per_step * (max(1, spp) / max(1, sum(step_spp) / max(1, args.steps)))
        pipe_gbs = pipe_bytes / (step_ms_of_one_step * 1e-3) / 1e9

        scene_info = scene.info()
        traffic, traffic_detail = None, {"skipped": "--no-live-traffic, a multi-GPU run or another configuration than c4"}
        if world == 1 and args.config == "c4" and not args.no_live_traffic:
            # free this process's queues and buffers first: the child renders the same batch on the same GPU
            scene.close()
            scene = None
            del acc
            torch.cuda.empty_cache()
            traffic, traffic_detail = live_extend_traffic(args)
        bytes_per_launch = ext_bytes / max(1, ext_launches)
        roofline = {"kernel": "k_extend", "bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved_gbs / HBM_PEAK_GBS, 5),
                    "traffic": round(traffic, 1) if traffic else None, "traffic_detail": traffic_detail,
                    "traffic_over_algorithmic": round(traffic / bytes_per_launch, 3) if traffic and spp == 2 * PMC_SPP else None,
                    "bytes_per_launch": round(bytes_per_launch, 1), "launches": ext_launches,
                    "avg_launch_ms": round(ext_ms / max(1, ext_launches), 5), "segments_per_sample": round(segments / max(1, cnt["samples"]), 4),
                    "measured": "HIP events around every launch of one more step after the timed region (same sample numbers per step, two pipelines)",
                    "alone": {"note": "one step with a single pipeline: k_extend does not share the chip with the other batch's kernels",
                              "avg_launch_ms": round(iso_ms / max(1, iso_launches), 5),
                              "achieved": round(ext_bytes / (iso_ms * 1e-3) / 1e9, 2) if iso_ms > 0 else 0.0,
                              "frac": round(ext_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if iso_ms > 0 else 0.0},
                    "valu": {"achieved": round(achieved_tf, 3), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(achieved_tf / FP32_VECTOR_PEAK_TFLOPS, 5)},
                    "pipeline": {"note": "the whole step against the HBM peak: (128 B x segments + 32 B x samples) / ms_per_step of the timed region",
                                 "bytes_per_step": pipe_bytes, "achieved": round(pipe_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(pipe_gbs / HBM_PEAK_GBS, 4),
                                 "algorithmic_bytes_per_segment": round(pipe_bytes / max(1, segments), 1)},
                    "work_per_segment": per_segment, "phase_ms": {k: round(v[0], 3) for k, v in phases.items()}}
        parallelism = f"sample-slice x{world}"
        if scaling == "weak":
            workload = (f"{job['name']} {w}x{h}, max depth {max_depth}, {spp} spp per step per GPU, sample-sliced over {world} GPU(s) of the "
                        f"{job['job_spp']}-spp job, one all-reduce of the accumulation buffer")
        else:
            workload = (f"{job['name']} {w}x{h}, max depth {max_depth}, the whole {job_samples_per_pixel}-spp job sample-sliced over {world} GPU(s) "
                        f"({count} spp on rank 0, in {args.steps} steps), one all-reduce of the accumulation buffer ({w * h * 16 / 1e6:.1f} MB)")
        metric = "Msamples/sec (whole node) on Book-2 final scene" if args.config == "c4" else "Msamples/sec (whole node) on Cornell box 1024x1024 10k spp"
        result = {
            "metric": metric, "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "width": w, "height": h, "spp_per_step": spp, "job_spp": job_samples_per_pixel,
                       "max_depth": max_depth, "parallelism": parallelism},
            "timed_region": {"seconds": round(elapsed, 5),
                             "per_rank_ms": [{"rank": i, "render": round(r, 3), "reduce_incl_wait_for_slowest_rank": round(x, 3)} for i, (r, x) in enumerate(per_rank)],
                             "allreduce_alone_ms": round(allreduce_alone_ms, 3) if allreduce_alone_ms is not None else None,
                             "allreduce_bytes": w * h * 16,
                             "note": "render: start of the timed region to this rank's last kernel; reduce: from there to the end of the all-reduce "
                                     "(includes waiting for the slowest rank); allreduce_alone: the same collective entered by all ranks at once, untimed region"},
            "roofline": roofline,
            "from_profiles": from_profiles(),
            "scene": scene_info,
        }
        if world == 1 and not args.no_configs and args.config == "c4":
            if scene is not None:
                scene.close()
                scene = None
                del acc
                torch.cuda.empty_cache()
            result["configs"] = other_configs(rt, local_rank)
        if not args.no_cpu_baseline and world == 1 and args.config == "c4":  # the host cores are timed on single-GPU runs only
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
