"""Developer diagnostic: per-ray GPU vs oracle agreement on primary + secondary rays."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as rt
import oracle_binding as ob

def report(name, gpu, cpu, rays):
    hm = gpu["hit"] == cpu["hit"]
    both = (gpu["hit"] == 1) & (cpu["hit"] == 1)
    same = both & (gpu["prim_id"] == cpu["prim_id"])
    err = np.abs(gpu["t"] - cpu["t"]) / np.maximum(1.0, np.abs(cpu["t"]))
    print(f"{name}: n={len(rays)} hit-agree={hm.mean():.6f} both={both.sum()} same-prim={same.sum()} "
          f"max relerr(same)={err[same].max() if same.any() else 0:.3e}")
    bad = np.where(same & (err > 1e-4))[0]
    for i in bad[:8]:
        print("   bad", i, "prim", gpu["prim_id"][i], "mat", gpu["material"][i], "t", gpu["t"][i], cpu["t"][i], "o", rays["origin"][i], "d", rays["direction"][i])
    diff = np.where(both & ~same)[0]
    for i in diff[:5]:
        print("   diffprim", i, gpu["prim_id"][i], cpu["prim_id"][i], gpu["t"][i], cpu["t"][i])
    nm = np.where(~hm)[0]
    for i in nm[:5]:
        print("   hit-mismatch", i, "gpu", gpu["hit"][i], gpu["t"][i], gpu["prim_id"][i], "cpu", cpu["hit"][i], cpu["t"][i], cpu["prim_id"][i])
    if same.any():
        ne = np.abs(gpu["normal"][same] - cpu["normal"][same]).max()
        pe = (np.abs(gpu["point"][same] - cpu["point"][same]).max(axis=1) / np.maximum(1, np.abs(cpu["point"][same]).max(axis=1))).max()
        print(f"   normal max abs err {ne:.3e}  point max rel err {pe:.3e}  ff agree {(gpu['front_face'][same]==cpu['front_face'][same]).mean():.6f} mat agree {(gpu['material'][same]==cpu['material'][same]).mean():.6f}")

for sid in [int(x) for x in sys.argv[1:]] or [8, 0, 6, 7, 1, 100]:
    w, h = 96, 64
    scene = rt.Scene.builtin(sid, w, h)
    oracle = ob.OracleScene.builtin(sid, w, h)
    pixels = np.arange(w * h, dtype=np.uint32)
    rays = scene.primary_rays(pixels, sample=1, seed=5)
    orays = oracle.primary_rays(pixels, sample=1, seed=5)
    for f in ("origin", "direction", "time"):
        d = np.abs(rays[f] - orays[f]).max()
        print(f"scene {sid} primary {f} max abs diff {d:.3e}")
    for bounce in range(4):
        rays["rng_bounce"] = bounce
        gpu = scene.intersect(rays, seed=5)
        cpu = oracle.intersect(rays, seed=5)
        report(f"scene {sid} bounce {bounce}", gpu, cpu, rays)
        # scatter with the oracle to get next rays (valid hits only)
        keep = cpu["hit"] == 1
        sc, att, emit = oracle.shade(rays[keep], cpu[keep], seed=5)
        gsc, gatt, gemit = scene.shade(rays[keep], cpu[keep], seed=5)
        ok = sc["tmin"] == gsc["tmin"]
        both_s = (sc["tmin"] == 1) & (gsc["tmin"] == 1)
        de = np.abs(sc["direction"][both_s] - gsc["direction"][both_s]).max() if both_s.any() else 0
        ae = np.abs(att - gatt).max(); ee = np.abs(emit - gemit).max()
        print(f"   shade: scatter-flag agree {ok.mean():.6f}, dir max abs err {de:.3e}, att err {ae:.3e}, emit err {ee:.3e}")
        nxt = sc[sc["tmin"] == 1].copy()
        if len(nxt) == 0: break
        src = rays[keep][sc["tmin"] == 1]
        nxt["tmin"] = 0.001; nxt["tmax"] = np.inf
        nxt["rng_pixel"] = src["rng_pixel"]; nxt["rng_sample"] = src["rng_sample"]
        rays = nxt
