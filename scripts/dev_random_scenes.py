"""More seeds of the random-scene parity test (tests/test_gpu_parity.py::test_random_scene_through_the_builder_api)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
for seed in [int(x) for x in os.environ.get("SEEDS", "1,2,3,4,5,6,8,9,10,11,12,13").split(",")]:
    try:
        T.test_random_scene_through_the_builder_api(seed)
        print("seed", seed, "ok", flush=True)
    except AssertionError as e:
        import traceback
        tb = traceback.extract_tb(e.__traceback__)
        print("seed", seed, "FAILED:", str(e)[:200], "|", "; ".join(f"{f.name}:{f.lineno} {f.line}" for f in tb[-2:]), flush=True)
