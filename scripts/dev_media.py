import os, sys, time
sys.path.insert(0, os.getcwd())
import raytracing_amd as rt
for sid in (7, 1):
    s = rt.Scene.builtin(sid, 1024, 1024, max_depth=25)
    s.render_samples(16); s.synchronize()
    t = time.time(); s.render_samples(16); s.synchronize(); dt = time.time() - t
    print(f"scene {sid}: {dt*1e3:.1f} ms")
