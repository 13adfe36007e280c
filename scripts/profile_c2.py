import os, sys
sys.path.insert(0, "/root/repo")
import raytracing_amd as rt
scene = rt.Scene.builtin(100, 1920, 1080, max_depth=25)
scene.set_pipelines(1)
for i in range(3):
    scene.render_samples(16)
scene.synchronize()
print("done", scene.last_render_ms())
