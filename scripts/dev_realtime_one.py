"""One interactive configuration in a loop (for a kernel trace of the frame's timeline): CASE=scene,width,height FRAMES=n"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt
sid, w, h = [int(x) for x in os.environ.get("CASE", "8,800,800").split(",")]
n = int(os.environ.get("FRAMES", "30"))
s = rt.Scene.builtin(sid, w, h)
for _ in range(10):
    s.realtime_render(copy=os.environ.get("COPY", "1") == "1")
t = time.time()
for _ in range(n):
    s.realtime_render(copy=os.environ.get("COPY", "1") == "1")
dt = (time.time() - t) / n
print(f"scene {sid} {w}x{h}: realtime_render {dt * 1e3:.3f} ms per frame", flush=True)
