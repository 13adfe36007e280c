"""Developer diagnostic: whole-image GPU vs oracle (counter RNG) and first timings."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as rt
import oracle_binding as ob

def enc(acc, n):
    return np.sqrt(np.clip(acc[..., :3] / n, 0, 1))

for sid, w, h, spp, depth in [(8, 160, 90, 8, 25), (0, 200, 112, 8, 50), (6, 128, 128, 8, 25), (7, 128, 128, 8, 25), (1, 160, 90, 8, 25), (100, 160, 90, 8, 25), (3, 160, 90, 8, 25)]:
    scene = rt.Scene.builtin(sid, w, h, max_depth=depth)
    oracle = ob.OracleScene.builtin(sid, w, h, max_depth=depth)
    scene.set_counters(True)
    t = time.time(); scene.render_samples(spp); g = scene.accumulator(); tg = time.time() - t
    gc = scene.counters(reset=True); scene.set_counters(False)
    from raytracing_amd import _abi
    oc = _abi.RtCounters()
    t = time.time(); c = oracle.render(spp, counters=oc); tc = time.time() - t
    a, b = enc(g, spp), enc(c, spp)
    rmse = np.sqrt(np.mean((a - b) ** 2))
    exact = np.mean(np.all(g[..., :3] == c[..., :3], axis=-1))
    close = np.mean(np.all(np.abs(g[..., :3] - c[..., :3]) <= 1e-4 * np.maximum(1, np.abs(c[..., :3])), axis=-1))
    print(f"scene {sid} {w}x{h}x{spp}: RMSE(sqrt-enc) {rmse:.5f}  pixels bit-equal {exact:.4f} close {close:.4f}  mean gpu {g[...,:3].mean()/spp:.5f} cpu {c[...,:3].mean()/spp:.5f}  gpu {tg:.2f}s cpu {tc:.2f}s")
    print("   gpu counters", gc)
    print("   cpu counters", oc.as_dict())
    if sid == 8:
        from PIL import Image
        Image.fromarray((a * 255.999).astype(np.uint8)).save(os.path.join(ROOT, "gpurun_out", "gpu_scene8.png"))

# timing at scale
for sid, w, h, spp, depth in [(8, 1920, 1080, 16, 25), (8, 3840, 2160, 4, 25), (0, 1600, 900, 16, 50), (6, 1024, 1024, 16, 25)]:
    scene = rt.Scene.builtin(sid, w, h, max_depth=depth)
    scene.render_samples(1); scene.synchronize()
    scene.set_phase_timing(True)
    t = time.time(); scene.render_samples(spp); scene.synchronize(); dt = time.time() - t
    print(f"scene {sid} {w}x{h}x{spp}: {dt:.3f}s wall, event {scene.last_render_ms():.1f} ms -> {w*h*spp/dt/1e6:.1f} Msamples/s; phases {scene.phase_ms()}")
