"""The PCIe-inclusive rate: rt_render_accumulate on a host buffer against the device-buffer variant (headline scene)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracing_amd as rt
w, h = 3840, 2160
scene = rt.Scene.builtin(8, w, h, max_depth=25, device=0)
import ctypes as C
host = np.zeros((h, w, 4), np.float32)
if os.environ.get("LOCKED", "0") == "1":   # the buffer in page-locked memory from rt_host_buffer_create
    p = C.c_void_p()
    scene._check(scene.amd.rt_host_buffer_create(scene.ctx, host.nbytes, C.byref(p)))
    host = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(h, w, 4))
    host[:] = 0
dev = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for spp in (1, 8, 32):
    scene.render_accumulate_host(host, 1, spp)
    t = time.perf_counter(); scene.render_accumulate_host(host, 1 + spp, spp); t_host = time.perf_counter() - t
    scene.render_accumulate_device(dev.data_ptr(), 1, spp); torch.cuda.synchronize()
    t = time.perf_counter(); scene.render_accumulate_device(dev.data_ptr(), 1 + spp, spp); torch.cuda.synchronize(); t_dev = time.perf_counter() - t
    print(f"{spp:3d} spp per call: host buffer {t_host * 1e3:7.1f} ms ({w * h * spp / t_host / 1e6:7.1f} Msamples/s), "
          f"device buffer {t_dev * 1e3:7.1f} ms ({w * h * spp / t_dev / 1e6:7.1f} Msamples/s), copies {1e3 * (t_host - t_dev):.1f} ms")
