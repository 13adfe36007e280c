"""Lane slots per phase of the extend kernel (development builds with RT_LANE_STATS, see rt_kernels.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
sid = int(os.environ.get("SCENE", "8"))
w, h, spp = 1920, 1080, 4
scene = rt.Scene.builtin(sid, w, h, max_depth=25, device=0)
acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
scene.set_counters(True)
scene.render_accumulate_device(acc.data_ptr(), 1, spp, seed=0)
torch.cuda.synchronize()
import ctypes as C
from raytracing_amd import _abi
c = _abi.RtCounters()
scene.amd.rt_ctx_get_counters(scene.ctx, C.byref(c), 1)
d = {k: int(getattr(c, k)) for k, _ in c._fields_}
print(d, flush=True)
