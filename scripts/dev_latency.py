import os, sys, time
sys.path.insert(0, os.getcwd())
import raytracing_amd as rt
for sid, w, h in [(8, 800, 800), (6, 1024, 1024)]:
    s = rt.Scene.builtin(sid, w, h)
    for _ in range(3): s.render_samples(1)
    s.synchronize()
    n = 30; te = 0.0; tt = 0.0
    for _ in range(n):
        t0 = time.perf_counter(); s.render_samples(1); t1 = time.perf_counter(); s.synchronize(); t2 = time.perf_counter()
        te += t1 - t0; tt += t2 - t0
    print(f"scene {sid} {w}x{h}: enqueue {te/n*1e3:.2f} ms, until done {tt/n*1e3:.2f} ms, GPU time of the render {s.last_render_ms():.2f} ms")
