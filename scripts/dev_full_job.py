"""The headline job in full on one GPU: Book-2 final scene 3840x2160, depth 25, 4096 samples per pixel
(the reference's --image mode: sample_count - 1 accumulate-only samples, then hdr_render), written as OpenEXR."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracing_amd as rt
spp = int(os.environ.get("SPP", "4096"))
s = rt.Scene.builtin(8, 3840, 2160, max_depth=25)
s.render_samples(1); s.synchronize()          # context, queues and scene are in place
s2 = rt.Scene.builtin(8, 3840, 2160, max_depth=25)
del s
t0 = time.time()
out = os.path.join(ROOT, "gpurun_out", "book2_final.exr")
os.makedirs(os.path.dirname(out), exist_ok=True)
img = s2.render_to_image(spp, out)
dt = time.time() - t0
print(f"{spp} spp of 3840x2160: {dt:.2f} s including scene build, EXR write ({os.path.getsize(out) / 1e6:.0f} MB) -> "
      f"{3840 * 2160 * spp / dt / 1e6:.0f} Msamples/s; image mean {img[..., :3].mean():.4f}, finite {bool(np.isfinite(img).all())}", flush=True)
os.remove(out)
