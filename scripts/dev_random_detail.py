"""The worst rays of the random-scene comparison for a few seeds (development aid for scripts/dev_random_scenes.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from test_gpu_parity import rt, ob, _abi
for seed in [int(x) for x in os.environ.get("SEEDS", "120,124,129").split(",")]:
    gpu = T._random_scene(rt.Scene(), np.random.default_rng(seed))
    cpu = T._random_scene(ob.OracleScene(), np.random.default_rng(seed))
    rng = np.random.default_rng(seed + 1)
    n = 20000
    rays = np.zeros(n, _abi.ray_dtype())
    rays["origin"] = rng.uniform(-8, 8, (n, 3))
    rays["direction"] = rng.normal(size=(n, 3)) * rng.uniform(0.1, 5, (n, 1))
    rays["time"] = rng.uniform(0, 1, n)
    rays["tmin"], rays["tmax"] = 0.001, np.inf
    rays["rng_pixel"], rays["rng_sample"], rays["rng_bounce"] = np.arange(n), 3, rng.integers(0, 8, n)
    want, got = cpu.intersect(rays, seed=99), gpu.intersect(rays, seed=99)
    both = (got["hit"] == 1) & (want["hit"] == 1) & (got["prim_id"] == want["prim_id"])
    terr = np.abs(got["t"] - want["t"]) / np.maximum(1.0, np.abs(want["t"]))
    perr = np.abs(got["point"] - want["point"]).max(axis=1) / np.maximum(1.0, np.abs(want["point"]).max(axis=1))
    bad = np.nonzero(both & ((terr > 1e-5) | (perr > 1e-4)))[0]
    print("seed", seed, "hit mismatch", int((got["hit"] != want["hit"]).sum()), "bad", len(bad))
    for i in bad[:5]:
        print("  ray", i, "o", rays["origin"][i], "d", rays["direction"][i], "bounce", rays["rng_bounce"][i])
        print("    want t", want["t"][i], "p", want["point"][i], "n", want["normal"][i], "prim", want["prim_id"][i], "mat", want["material"][i])
        print("    got  t", got["t"][i], "p", got["point"][i], "n", got["normal"][i], "prim", got["prim_id"][i], "mat", got["material"][i])
