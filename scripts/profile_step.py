"""A few steps of the benchmark workload without torch, for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt
w, h, spp, steps = 3840, 2160, int(os.environ.get("SPP", "4")), int(os.environ.get("STEPS", "2"))
scene = rt.Scene.builtin(8, w, h, max_depth=25)
scene.set_pipelines(int(os.environ.get("PIPES", "2")))
for i in range(steps):
    scene.render_samples(spp)
scene.synchronize()
print("done", scene.last_render_ms(), "ms for the last step")
