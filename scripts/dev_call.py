"""Wall-clock time per render call on a device buffer (development aid): SCENE, W, H, DEPTH, SPP from the environment."""
import os, sys, time
sys.path.insert(0, '.')
import torch, raytracing_amd as rt
sid, w, h = int(os.environ.get("SCENE", "0")), int(os.environ.get("W", "400")), int(os.environ.get("H", "225"))
depth, spp = int(os.environ.get("DEPTH", "50")), int(os.environ.get("SPP", "100"))
s = rt.Scene.builtin(sid, w, h, max_depth=depth)
acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for i in range(4): s.render_accumulate_device(acc.data_ptr(), 1 + spp * i, spp, seed=0)
torch.cuda.synchronize(); s.synchronize()
n = max(3, min(40, int(2e9 / (w * h * spp * 25))))
t0 = time.perf_counter()
for i in range(n): s.render_accumulate_device(acc.data_ptr(), 1000 + spp * i, spp, seed=0)
torch.cuda.synchronize(); s.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"scene {sid} {w}x{h} depth {depth} spp {spp}: {w*h*spp/1e6:.1f} M paths, {dt*1e3:.3f} ms per call, {w*h*spp/dt/1e6:.0f} Msamples/s (pipes {os.environ.get('RT_AMD_PIPES')})")
