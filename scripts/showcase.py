"""Look-alike check against the only rendered outputs the reference holds: showcases/rgb0..8.png.

The PNGs were made at unknown sample counts, by unknown (in places older) revisions of the reference, and with an
unknown output mapping, so this is a qualitative check - it cannot pin parity - but it is the one artefact that
would show a slip shared by this repository's two transcriptions of the reference's scene builders
(raytracing_amd/host/scenes.cpp and oracle/oracle_scenes.hpp, both from src/main.rs:291-731).

    python scripts/showcase.py render  [out_dir]     on the GPU box: scenes 0-3, 5-8 at the PNGs' sizes, SPP samples,
                                                     mean radiance block-averaged 4x4 -> out_dir/sceneN.npy
    python scripts/showcase.py compare [out_dir]     where /root/reference exists: per-scene RMSE and mean-colour
                                                     deltas of both output mappings of the reference against rgbN.png
Scene 4 is left out: the reference's builder opens earthmap3.png, which it does not ship.
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# scene id -> (showcase PNG, its size): the PNGs are numbered as the scenes of src/main.rs:137-151 are, except that rgb1.png
# shows the five quads (scene 2) and rgb2.png the night variant of the spheres (scene 1)
PNG_OF = {0: 0, 1: 2, 2: 1, 3: 3, 5: 5, 6: 6, 7: 7, 8: 8}
SIZES = {0: (1920, 1080), 1: (1920, 1080), 2: (1200, 1200), 3: (1920, 1080), 5: (1920, 1200), 6: (1200, 1200), 7: (1200, 1200), 8: (1200, 1200)}
SPP = int(os.environ.get("SPP", "256"))
BLOCK = 4


def block_mean(img):
    h, w = img.shape[0] // BLOCK * BLOCK, img.shape[1] // BLOCK * BLOCK
    return img[:h, :w].reshape(h // BLOCK, BLOCK, w // BLOCK, BLOCK, -1).mean(axis=(1, 3))


def render(out_dir):
    import raytracing_amd as rt
    os.makedirs(out_dir, exist_ok=True)
    for sid, (w, h) in SIZES.items():
        s = rt.Scene.builtin(sid, w, h)
        s.render_samples(SPP)
        mean = s.accumulator()[..., :3] / SPP
        np.save(os.path.join(out_dir, f"scene{sid}.npy"), block_mean(mean).astype(np.float32))
        print(f"scene {sid}: {w}x{h}, {SPP} spp, mean radiance {mean.mean():.4f}", flush=True)
        s.close()


def compare(out_dir):
    from PIL import Image
    # The reference's two per-pixel mappings: hdr_render's sqrt(mean) (src/camera.rs:115-118), and realtime_render's
    # 1 - exp(-exposure c) followed by to_rgba's sqrt (src/camera.rs:95-100, src/math/vec3.rs:253-259).  The viewer's
    # exposure is changed with the arrow keys (src/main.rs:240-262) and the value behind a PNG is not recorded, so
    # the exposure that brings the render closest is fitted per image and reported.
    exposures = np.exp(np.linspace(np.log(0.125), np.log(32.0), 65))
    print("| scene | PNG | RMSE, sqrt(mean) | RMSE, tonemap at the fitted exposure | fitted exposure | mean colour ours (r, g, b) | mean colour PNG | correlation of the luminance images |")
    print("|---|---|---|---|---|---|---|---|")
    for sid, (w, h) in SIZES.items():
        png = np.asarray(Image.open(f"/root/reference/showcases/rgb{PNG_OF[sid]}.png").convert("RGB"), np.float32) / 255.0
        assert png.shape[:2] == (h, w), (sid, png.shape)
        ref = block_mean(png)
        mean = np.load(os.path.join(out_dir, f"scene{sid}.npy"))
        # mapping the block mean is not the block mean of the mapped image; at 256 spp and 4x4 blocks the difference is
        # small against the differences looked for here (wrong colours, missing objects, shifted geometry)
        rmse = lambda img: float(np.sqrt(np.mean((img - ref) ** 2)))
        plain = rmse(np.sqrt(np.clip(mean, 0, 1)))
        tone = [rmse(np.sqrt(1.0 - np.exp(-e * np.clip(mean, 0, None)))) for e in exposures]
        k = int(np.argmin(tone))
        ours_img = np.sqrt(1.0 - np.exp(-exposures[k] * np.clip(mean, 0, None)))
        ours, theirs = ours_img.mean(axis=(0, 1)), ref.mean(axis=(0, 1))
        corr = float(np.corrcoef(ours_img.mean(axis=2).ravel(), ref.mean(axis=2).ravel())[0, 1])
        print(f"| {sid} | rgb{PNG_OF[sid]}.png {w}x{h} | {plain:.3f} | {tone[k]:.3f} | {exposures[k]:.2f} | {ours[0]:.3f}, {ours[1]:.3f}, {ours[2]:.3f} | "
              f"{theirs[0]:.3f}, {theirs[1]:.3f}, {theirs[2]:.3f} | {corr:.3f} |")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "render"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "showcase")
    (render if mode == "render" else compare)(out)
