"""The benchmark configurations other than the headline one, timed on one GPU (wall clock of whole render calls,
scene build reported separately).  Usage: dev_configs.py [c1] [c2] [c3] [c5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt

CONFIGS = {
    # name: (scene id, width, height, depth, spp per call, calls, extra)
    "c1": (0, 400, 225, 50, 100, 3, {}),                       # Book-1 random spheres, the reference's own CPU case
    "c2": (100, 1920, 1080, 25, 50, 3, {}),                    # earth + perlin spheres
    "c3": (6, 1024, 1024, 25, 64, 3, {}),                      # Cornell box
    "c5": (102, 1920, 1080, 2, 16, 3, {"stress_count": int(os.environ.get("STRESS", "10000000"))}),
}
for name in (sys.argv[1:] or list(CONFIGS)):
    sid, w, h, depth, spp, calls, extra = CONFIGS[name]
    t0 = time.time()
    scene = rt.Scene.builtin(sid, w, h, max_depth=depth, **extra)
    t_host = time.time() - t0
    scene.render_samples(1)
    scene.synchronize()
    info = scene.info()
    scene.set_counters(True)
    scene.render_samples(1)
    scene.synchronize()
    cnt = scene.counters(reset=True)
    scene.set_counters(False)
    scene.render_samples(spp)
    scene.synchronize()
    t0 = time.time()
    for _ in range(calls):
        scene.render_samples(spp)
    scene.synchronize()
    dt = (time.time() - t0) / calls
    seg = cnt["segments"] / cnt["samples"]
    print(f"{name}: scene {sid} {w}x{h} depth {depth}, {spp} spp per call: {dt * 1e3:.1f} ms -> {w * h * spp / dt / 1e6:.1f} Msamples/s, "
          f"{w * h * spp * seg / dt / 1e6:.0f} Mrays/s ({seg:.2f} segments per sample, {cnt['node_visits'] / cnt['segments']:.2f} node visits, {cnt['sphere_tests'] / cnt['segments']:.2f} sphere tests, "
          f"{cnt['quad_tests'] / cnt['segments']:.2f} quad tests per segment, deepest stack {cnt['max_stack']}); "
          f"host scene build {t_host:.2f} s, BVH {info['bvh_nodes']} nodes depth {info['bvh_depth']} in {info['build_seconds']:.2f} s, "
          f"{info['device_bytes'] / 1e6:.1f} MB on the GPU", flush=True)
