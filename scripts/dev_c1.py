import os, sys, time
sys.path.insert(0, '.')
import torch, raytracing_amd as rt
s = rt.Scene.builtin(0, 400, 225, max_depth=50)
acc = torch.zeros((225, 400, 4), dtype=torch.float32, device="cuda")
for i in range(5): s.render_accumulate_device(acc.data_ptr(), 1 + 100 * i, 100, seed=0)
torch.cuda.synchronize(); s.synchronize()
t0 = time.perf_counter()
n = 40
for i in range(n): s.render_accumulate_device(acc.data_ptr(), 1000 + 100 * i, 100, seed=0)
torch.cuda.synchronize(); s.synchronize()
print(os.environ.get("RT_AMD_TAIL_BLOCKS_PER_CU"), os.environ.get("RT_AMD_LIBRARY", "")[-12:], "C1 ms per call", (time.perf_counter() - t0) / n * 1e3)
