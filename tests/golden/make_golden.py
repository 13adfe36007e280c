"""Regenerates the golden fixtures in this directory from the CPU oracle.

    python tests/golden/make_golden.py

The reference ships no golden vectors for its render path and cannot be built in
this image (nightly Rust + un-vendored crates), so these fixtures pin the
*oracle's* behaviour: known answers that were checked by hand or against an
independent Python restatement (see tests/test_oracle_kat.py), ray batches with
the oracle's hit records, and small low-spp images.  They guard the oracle
against drift and give the GPU tests data that does not depend on rebuilding the
oracle.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_binding as ob  # noqa: E402

SCENES = {0: (48, 27, 50), 1: (48, 27, 25), 3: (48, 27, 25), 6: (32, 32, 25), 7: (32, 32, 25), 8: (32, 32, 25), 100: (48, 27, 25),
          101: (32, 32, 25)}
SPP = 4
SEED = 7


def main():
    kat = {
        "xoshiro_seed": [69, 420, 420, 69],
        "xoshiro_next": [str(x) for x in ob.seed_next([69, 420, 420, 69], 8)],
        "xoshiro_next_f32": [float.hex(x) for x in ob.seed_next_f32([69, 420, 420, 69], 8)],
        "row_seed_first_f32": float.hex(ob.seed_next_f32([0, 1, 90000, 0], 1)[0]),
    }
    u, f = ob.counter_draws(SEED, 12345, 17, 3, 1, 8)
    kat["counter"] = {"seed": SEED, "pixel": 12345, "sample": 17, "bounce": 3, "stream": 1, "u32": [int(x) for x in u],
                      "f32": [float.hex(float(x)) for x in f]}
    with open(os.path.join(HERE, "kat.json"), "w") as fh:
        json.dump(kat, fh, indent=1)

    for sid, (w, h, depth) in SCENES.items():
        o = ob.OracleScene.builtin(sid, w, h, max_depth=depth)
        pixels = np.arange(w * h, dtype=np.uint32)
        rays = o.primary_rays(pixels, sample=1, seed=SEED)
        hits = o.intersect(rays, seed=SEED)
        # one scattered generation, so that rays start on surfaces and inside media too
        keep = hits["hit"] == 1
        sc, att, emit = o.shade(rays[keep], hits[keep], seed=SEED)
        ok = sc["tmin"] == 1
        rays2 = sc[ok].copy()
        rays2["tmin"] = 0.001
        rays2["tmax"] = np.inf
        rays2["rng_pixel"] = rays[keep][ok]["rng_pixel"]
        rays2["rng_sample"] = 1
        rays2["rng_bounce"] = 1
        hits2 = o.intersect(rays2, seed=SEED)
        img_counter = o.render(SPP, mode=ob.RNG_COUNTER, seed=SEED, threads=4)
        img_xoshiro = o.render(SPP, mode=ob.RNG_XOSHIRO, threads=4)
        np.savez_compressed(os.path.join(HERE, f"scene{sid}.npz"), width=w, height=h, max_depth=depth, spp=SPP, seed=SEED,
                            rays=rays, hits=hits, shade_scattered=sc, shade_attenuation=att, shade_emitted=emit, rays2=rays2,
                            hits2=hits2, acc_counter=img_counter, acc_xoshiro=img_xoshiro)
        print(f"scene {sid}: {len(rays)} + {len(rays2)} rays, image mean {img_counter[..., :3].mean() / SPP:.4f}")


if __name__ == "__main__":
    main()
