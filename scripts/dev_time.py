"""Developer timing: a few configurations, wall + per-phase HIP-event times."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt
SPP8 = int(os.environ.get("SPP8", "8"))
cfgs = [(8, 3840, 2160, SPP8, 25), (0, 1600, 900, 16, 50), (6, 1024, 1024, 16, 25), (100, 1920, 1080, 16, 25)]
if len(sys.argv) > 1: cfgs = cfgs[:int(sys.argv[1])]
if os.environ.get("ONLY"): cfgs = [c for c in cfgs if str(c[0]) in os.environ["ONLY"].split()]
for sid, w, h, spp, depth in cfgs:
    scene = rt.Scene.builtin(sid, w, h, max_depth=depth)
    if os.environ.get("BATCH_MI"): scene.set_batch_paths(int(os.environ["BATCH_MI"]) << 20)
    scene.render_samples(spp); scene.synchronize()
    scene.set_phase_timing(True)
    t = time.time(); scene.render_samples(spp); scene.synchronize(); dt = time.time() - t
    ph = {k: round(v[0], 2) for k, v in scene.phase_ms().items()}
    print(f"scene {sid} {w}x{h}x{spp}: {dt*1e3:.1f} ms -> {w*h*spp/dt/1e6:.1f} Msamples/s; phases {ph}", flush=True)
