"""Life spans of the extend kernel's waves on one scene (development aid; needs a build with -DRT_WAVE_TIMES, which prints a
line per 16th wave).  usage: RT_AMD_LIBRARY=variants/librt_amd_wt.so python scripts/wave_times.py > log; python scripts/wave_times.py --parse log"""
import os, sys, re
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import numpy as np
    rows = [tuple(int(x) for x in re.findall(r"\d+", l)) for l in open(sys.argv[2]) if l.startswith("wave ")]
    rows.sort(key=lambda r: r[2])
    groups, cur = [], []
    for r in rows:  # a dispatch: starts within 2 ms of each other (100 MHz clock)
        if cur and (r[2] - cur[0][2] > 200000 or r[0] != cur[0][0]):
            groups.append(cur); cur = []
        cur.append(r)
    if cur: groups.append(cur)
    for g in groups:
        a = np.array(g, dtype=np.int64)
        t0 = a[:, 2].min()
        end = (a[:, 3] - t0) / 1e5; start = (a[:, 2] - t0) / 1e5
        turns = a[:, 4]; ovf = a[:, 5]
        q = lambda v: " ".join(f"{x:.2f}" for x in np.percentile(v, [0, 10, 50, 90, 99, 100]))
        print(f"bounce {g[0][0]}: {len(g)} waves sampled; start ms [{q(start)}]; end ms [{q(end)}]; turns [{q(turns)}]; overflow turns [{q(ovf)}]; "
              f"mean alive fraction {((a[:, 3] - a[:, 2]).mean() / 1e5) / end.max():.3f}; us per turn {((a[:, 3] - a[:, 2]) / np.maximum(turns, 1)).mean() / 100:.3f}")
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("RT_PKG_ROOT", ROOT))
import raytracing_amd as rt
sid = int(os.environ.get("SCENE", "102")); w = int(os.environ.get("W", "1920")); h = int(os.environ.get("H", "1080"))
extra = {"stress_count": int(os.environ.get("STRESS", "10000000"))} if sid == 102 else {}
scene = rt.Scene.builtin(sid, w, h, max_depth=int(os.environ.get("DEPTH", "2")), **extra)
scene.set_pipelines(1)
scene.render_samples(int(os.environ.get("SPP", "16")))
scene.synchronize()
