"""Random scenes made of big leaves (spheres, media bounded by spheres, moving spheres and boxes, in random order, some of
them sharing their sphere) over a few small ones: the leaves that are tested ahead of the tree, per ray against the oracle.
SEEDS=a,b,c selects the seeds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob
import raytracing_amd as rt
from raytracing_amd import _abi


def scene(s, rng):
    t = s.tex_solid(0.6, 0.6, 0.7)
    lam, glass = s.lambertian(t), s.dielectric(t, 1.5)
    items = []
    scale = float(10.0 ** rng.uniform(-1, 3))
    spheres = []
    for _ in range(int(rng.integers(1, 7))):
        kind = int(rng.integers(0, 6))
        c = tuple(rng.uniform(-0.3, 0.3, 3) * scale)
        r = float(rng.uniform(0.6, 1.2) * scale)
        if kind == 0:
            items.append(s.sphere(c, r, glass)); spheres.append((c, r))
        elif kind == 1 and spheres:      # a medium bounded by an earlier sphere (possibly the one just before it)
            c, r = spheres[int(rng.integers(len(spheres)))]
            items.append(s.medium(s.sphere(c, r, lam), float(rng.uniform(0.001, 0.05) / scale * 10), t))
        elif kind == 2:
            items.append(s.sphere(c, r, glass)); spheres.append((c, r))
            items.append(s.medium(s.sphere(c, r, glass), float(rng.uniform(0.001, 0.05) / scale * 10), t))
        elif kind == 3:
            items.append(s.medium(s.sphere(c, r, lam), float(rng.uniform(0.001, 0.05) / scale * 10), t))
        elif kind == 4:
            items.append(s.medium(s.box(tuple(-r for _ in range(3)), (r, r, r), lam), float(rng.uniform(0.001, 0.05) / scale * 10), t))
        else:
            items.append(s.medium(s.moving_sphere(c, tuple(np.array(c) + 0.1 * scale), r, lam), float(rng.uniform(0.001, 0.05) / scale * 10), t))
    for i in range(int(rng.integers(4, 12))):
        items.append(s.sphere(tuple(rng.uniform(-0.5, 0.5, 3) * scale), float(rng.uniform(0.01, 0.1) * scale), lam))
    s.set_world(s.bvh(items))
    s.camera((0, 0, -0.4 * scale), 32, 32, max_depth=4, vfov=60.0, background=(0.7, 0.8, 0.9), yaw=90.0)
    return s, scale


bad = 0
for seed in [int(x) for x in os.environ.get("SEEDS", ",".join(str(i) for i in range(40))).split(",")]:
    (gpu, scale), (cpu, _) = scene(rt.Scene(), np.random.default_rng(seed)), scene(ob.OracleScene(), np.random.default_rng(seed))
    rng = np.random.default_rng(seed + 1000)
    n = 20000
    rays = np.zeros(n, _abi.ray_dtype())
    rays["origin"] = rng.uniform(-1.4, 1.4, (n, 3)) * scale
    rays["direction"] = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 3, (n, 1))
    rays["time"] = rng.uniform(0, 1, n)
    rays["tmin"], rays["tmax"] = 0.001, np.inf
    rays["rng_pixel"], rays["rng_sample"], rays["rng_bounce"] = np.arange(n) % 4096, 1 + np.arange(n) // 4096, rng.integers(0, 4, n)
    want = cpu.intersect(rays, seed=seed)
    a, b = gpu.intersect(rays, seed=seed), gpu.extend(rays, seed=seed)
    hit = want["hit"] == 1
    ok = np.array_equal(a["hit"], want["hit"]) and np.array_equal(b["hit"], want["hit"])
    ok_id = ok and np.array_equal(a["prim_id"][hit], want["prim_id"][hit]) and np.array_equal(b["prim_id"][hit], want["prim_id"][hit])
    with np.errstate(all="ignore"):
        rel = np.abs(a["t"][hit] - want["t"][hit]) / np.maximum(np.abs(want["t"][hit]), 1e-30)
        same_ab = a["t"].tobytes() == b["t"].tobytes()
    n_exact = int((a["t"][hit] == want["t"][hit]).sum())
    info = gpu.info()
    good = ok and ok_id and same_ab and (rel.max(initial=0) <= 2e-5)
    bad += not good
    print(f"seed {seed:3d} scale {scale:9.3g} first {info['leaves_tested_first']} tree {info['leaves_in_tree']} hit {hit.mean():.2f} "
          f"hits_equal {ok} ids {ok_id} a==b {same_ab} exact_t {n_exact}/{int(hit.sum())} max_rel {rel.max(initial=0):.2e} {'ok' if good else 'FAILED'}", flush=True)
    if not good and os.environ.get("DETAIL"):
        idx = np.nonzero((a["hit"] != want["hit"]) | ((want["hit"] == 1) & ((a["prim_id"] != want["prim_id"]) | (np.abs(a["t"] - want["t"]) > 2e-5 * np.abs(want["t"])))))[0]
        print("  mismatches:", len(idx))
        for i in idx[:6]:
            print("  ray", i, "o", rays["origin"][i], "d", rays["direction"][i], "time", rays["time"][i], "bounce", rays["rng_bounce"][i])
            print("    want hit", want["hit"][i], "prim", want["prim_id"][i], "t", want["t"][i], "mat", want["material"][i], "| got hit", a["hit"][i], "prim", a["prim_id"][i], "t", a["t"][i], "mat", a["material"][i])
print("failures:", bad)
sys.exit(1 if bad else 0)
