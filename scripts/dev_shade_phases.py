"""Where a tile of k_shade spends its time.  Needs a development build of the library whose k_shade adds the 100 MHz
clock between its barriers into the work counters (not kept in the tree; see the kernel's comment block for the
phases)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
import raytracing_amd as rt
from raytracing_amd import _abi
sid = int(os.environ.get("SCENE", "8"))
w, h, spp = (3840, 2160, 8) if sid == 8 else (1024, 1024, 16)
scene = rt.Scene.builtin(sid, w, h, max_depth=25, device=0)
scene.set_pipelines(1)
acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
scene.render_accumulate_device(acc.data_ptr(), 1, spp, seed=0); torch.cuda.synchronize()
raw = _abi.RtCounters(); scene.amd.rt_ctx_get_counters(scene.ctx, C.byref(raw), 1)
scene.set_phase_timing(True)
scene.render_accumulate_device(acc.data_ptr(), 1, spp, seed=0); torch.cuda.synchronize(); scene.synchronize()
scene.amd.rt_ctx_get_counters(scene.ctx, C.byref(raw), 1)
names = ["classify (ballots .. barrier)", "prefix + scatter of the order", "shade rounds", "prefetch + barrier", "atomics + barrier", "copy-out + barrier"]
vals = [raw.node_visits, raw.sphere_tests, raw.quad_tests, raw.medium_tests, raw.segments, raw.hits]
tot = sum(vals)
print("scene", sid, "shade phase ms", scene.phase_ms()["shade"])
for n, v in zip(names, vals): print(f"  {n:32s} {100.0 * v / tot:5.1f} %")
