"""Latency of the interactive entry point (Camera::realtime_render: one more sample, tone-map, pack, download)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt
for sid, w, h in [(8, 800, 800), (8, 1920, 1080), (6, 1024, 1024), (0, 1600, 900)]:
    s = rt.Scene.builtin(sid, w, h)
    for _ in range(3):
        s.realtime_render()
    t = time.time()
    n = 20
    for _ in range(n):
        img = s.realtime_render(copy=os.environ.get("COPY", "1") == "1")
    dt = (time.time() - t) / n
    print(f"scene {sid} {w}x{h}: realtime_render {dt * 1e3:.2f} ms per frame ({1 / dt:.0f} frames/s, {w * h / dt / 1e6:.0f} Msamples/s)", flush=True)
