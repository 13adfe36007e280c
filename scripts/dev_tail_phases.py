"""Per-phase GPU time (HIP events around every launch) of one small frame for several tail starts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt
sid, w, h, depth = [int(x) for x in os.environ.get("CASE", "8,800,800,25").split(",")]
spp = int(os.environ.get("SPP", "1"))
s = rt.Scene.builtin(sid, w, h, max_depth=depth)
for st in [int(x) for x in os.environ.get("TAIL_STARTS", "1000,0,4,8").split(",")]:
    s.set_tail_start(st)
    s.render_samples(spp); s.render_samples(spp); s.synchronize()
    s.set_phase_timing(True)
    for _ in range(5):
        s.render_samples(spp)
    s.synchronize()
    ph = s.phase_ms()
    s.set_phase_timing(False)
    print(f"scene {sid} {w}x{h} tail start {st}: " + "  ".join(f"{k} {v[0] / 5:.3f} ms/{v[1] // 5}" for k, v in ph.items()), flush=True)
