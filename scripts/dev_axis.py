"""Axis-parallel rays from coordinates on the scene's planes (the rays of test_axis_parallel_rays_and_origins_on_box_planes):
where do the GPU kernels and the CPU restatement disagree?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracing_amd as rt
from raytracing_amd import _abi
import oracle_binding as ob
sid = int(os.environ.get("SCENE", "6"))
gpu, cpu = rt.Scene.builtin(sid, 16, 16, max_depth=4), ob.OracleScene.builtin(sid, 16, 16, max_depth=4)
cornell = [0.0, 555.0, 278.0, 130.0, 265.0, 295.0, 165.0, 330.0, 65.0, 460.0, 227.0, 332.0, 343.0, 213.0, 554.0]
planes = np.array({6: cornell, 7: cornell,
                   8: [0.0, 100.0, 200.0, 123.0, 423.0, 554.0, 147.0, 412.0, 278.0, -1000.0, 1000.0, 260.0, 150.0, 45.0],
                   0: [0.0, 1.0, -1000.0, 4.0, -4.0, 0.2, 2.0, 1000.0, -11.0, 11.0]}[sid], np.float32)
rng = np.random.default_rng(17 + sid)
n = 8000
dirs = rng.integers(-1, 2, (n, 3)).astype(np.float32)
dirs[(dirs == 0).all(axis=1)] = (0, 0, 1)
dirs *= rng.choice(np.array([1.0, 0.5, 3.0], np.float32), (n, 1))
dirs = np.where((dirs == 0) & (rng.random((n, 3)) < 0.25), np.float32(-0.0), dirs)
origin = rng.choice(planes, (n, 3))
off = rng.random((n, 3)) < 0.5
origin = np.where(off, origin + rng.uniform(-50, 50, (n, 3)).astype(np.float32), origin).astype(np.float32)
rays = np.zeros(n, _abi.ray_dtype())
rays["origin"], rays["direction"] = origin, dirs
rays["time"] = rng.uniform(0, 1, n)
rays["tmin"], rays["tmax"] = 0.001, np.inf
rays["rng_pixel"], rays["rng_sample"] = np.arange(n) % 256, 1 + np.arange(n) // 256
in_plane = ((dirs == 0) & np.isin(origin, planes)).any(axis=1)
want = cpu.intersect(rays)
got = gpu.extend(rays)
keep = ~in_plane
bad = np.where(keep & ((got["hit"] != want["hit"]) | ((got["hit"] == 1) & (got["prim_id"] != want["prim_id"]))))[0]
tdiff = np.where(keep & (got["hit"] == 1) & (want["hit"] == 1) & (got["prim_id"] == want["prim_id"]) & (got["t"] != want["t"]))[0]
print("rays outside every plane:", int(keep.sum()), "hit / primitive disagreements:", len(bad), "; same primitive, other t:", len(tdiff))
for i in tdiff[:10]:
    print("  T o", rays["origin"][i].tolist(), "d", rays["direction"][i].tolist(), "| gpu t", float(got["t"][i]), "prim", got["prim_id"][i], "| cpu t", float(want["t"][i]))
for i in bad[:6]:
    print("  H o", rays["origin"][i].tolist(), "d", rays["direction"][i].tolist(), "| gpu", got["hit"][i], float(got["t"][i]), got["prim_id"][i], "| cpu", want["hit"][i], float(want["t"][i]), want["prim_id"][i])
