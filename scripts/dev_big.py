"""Robustness runs: an image with more pixels than one batch holds (pixel chunks), and deep paths at 4K."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_amd as rt
# more pixels than one batch holds: the image is rendered in pixel chunks
s = rt.Scene.builtin(6, 16384, 8200, max_depth=8)
t = time.time(); s.render_samples(2); s.synchronize(); print("134M-pixel image, 2 spp:", round(time.time() - t, 2), "s")
acc = s.accumulator()
print("mean", acc[..., :3].mean(), "finite", np.isfinite(acc).all(), "rows nonzero", (acc[..., :3].sum(axis=(1, 2)) > 0).mean())
s2 = rt.Scene.builtin(0, 3840, 2160, max_depth=50)
t = time.time(); s2.render_samples(8); s2.synchronize(); print("scene 0 4K depth 50, 8 spp:", round(time.time() - t, 3), "s")
