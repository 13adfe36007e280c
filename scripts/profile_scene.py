"""One scene as a workload for `rocprofv3 --kernel-trace --stats`: SCENE, W, H, SPP, DEPTH from the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt
sid = int(os.environ.get("SCENE", "100"))
w, h, spp, depth = (int(os.environ.get(k, d)) for k, d in (("W", "1920"), ("H", "1080"), ("SPP", "16"), ("DEPTH", "25")))
scene = rt.Scene.builtin(sid, w, h, max_depth=depth, stress_count=int(os.environ.get("STRESS", "100000")))
scene.set_pipelines(1)
if "RT_TAIL" in os.environ:
    scene.set_tail_start(int(os.environ["RT_TAIL"]))
for i in range(3):
    scene.render_samples(spp)
scene.synchronize()
print("done", scene.last_render_ms())
