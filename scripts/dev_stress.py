import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as rt
import oracle_binding as ob
n = 200000
s = rt.Scene.builtin(102, 96, 54, max_depth=2, stress_count=n)
o = ob.OracleScene.builtin(102, 96, 54, max_depth=2, stress_count=n)
pix = np.arange(96 * 54, dtype=np.uint32)
rays = s.primary_rays(pix, 1)
g = s.intersect(rays); c = o.intersect(rays)
bad = np.where(g["hit"] != c["hit"])[0]
print("mismatch", bad)
L = o.leaves()
for i in bad:
    print("ray", rays[i]); print("gpu", g[i]); print("cpu", c[i])
    pid = c[i]["prim_id"] if c[i]["hit"] else g[i]["prim_id"]
    leaf = L[L["prim_id"] == pid][0]
    print("leaf", leaf["data"][:4])
    # double precision check
    oo = rays[i]["origin"].astype(np.float64); d = rays[i]["direction"].astype(np.float64)
    cc = leaf["data"][:3].astype(np.float64); r = float(leaf["data"][3])
    oc = oo - cc; a = d @ d; hb = oc @ d; cq = oc @ oc - r * r
    disc = hb * hb - a * cq
    print("float64 disc", disc, "t", (-hb - np.sqrt(max(disc, 0))) / a)
    # closest approach distance vs radius
    tca = -hb / a; dist = np.linalg.norm(oc + tca * d)
    print("closest approach", dist, "radius", r, "diff", dist - r)
