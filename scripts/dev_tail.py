"""Per-call time of small and mid-size batches with the tail kernel taking over at different bounces
(TAIL_STARTS: comma-separated first bounces, -1 = the default schedule, 1000 = per-bounce kernels only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt
starts = [int(x) for x in os.environ.get("TAIL_STARTS", "1000,-1,0,2,4,6,8,12").split(",")]
CASES = [(8, 800, 800, 25, 1), (8, 1920, 1080, 25, 1), (6, 1024, 1024, 25, 1), (0, 1600, 900, 25, 1), (0, 400, 225, 50, 100), (6, 1024, 1024, 25, 8)]
for sid, w, h, depth, spp in CASES:
    s = rt.Scene.builtin(sid, w, h, max_depth=depth)
    line = []
    for st in starts:
        s.set_tail_start(st)
        for _ in range(2):
            s.render_samples(spp)
        s.synchronize()
        n = 10
        t = time.time()
        for _ in range(n):
            s.render_samples(spp)
        s.synchronize()
        line.append(f"{st}: {(time.time() - t) / n * 1e3:.2f}")
    print(f"scene {sid} {w}x{h} depth {depth} {spp} spp/call, ms per call by tail start -> " + "  ".join(line), flush=True)
