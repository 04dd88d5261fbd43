"""Soak: many camera poses, every frame rendered under the block plan (serial and overlapped emission) and the
sort plan; sorted keys / values (position-weighted checksums), tile ranges and pixels must agree bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W, H = 1920, 1080
scene = scenes.garden_like_scene(1_500_000, seed=91)
span = float(np.max(scene["means3D"][:, :3].max(0) - scene["means3D"][:, :3].min(0)))
r = SplatRasterizer(W, H, background=(0.1, 0.2, 0.3))
r.configure_from_scene(scene)
rng = np.random.default_rng(5)


def digest():
    b = r.map_binning_state()
    k, v = b["keys"], b["values"].to(torch.int64)
    w = torch.arange(k.numel(), device=k.device, dtype=torch.int64) * 0x9E3779B1 + 12345
    return (int((k ^ w).sum()), int((v * (w | 1)).sum()), int(r.map_image_state()["ranges"].to(torch.int64).sum()))


bad = 0
for i in range(poses):
    pos = tuple(float(x) for x in rng.uniform(-6, 6, 3))
    cam = camera.first_person_camera(pos, float(rng.uniform(-3.14, 3.14)), float(rng.uniform(-0.8, 0.8)), float(np.radians(45.0)),
                                     0.001 * span, span, W, H, True)
    img_b = r.draw(cam, plan="blocks").clone()
    R = r.last_num_rendered
    if R == 0:
        continue
    d_b = digest()
    img_o = r.draw(cam, plan="blocks", overlap_emit=True).clone()
    d_o = digest()
    img_s = r.draw(cam, plan="sort")
    d_s = digest()
    ok = d_b == d_o == d_s and torch.equal(img_b, img_o) and torch.equal(img_b, img_s)
    bad += 0 if ok else 1
    if not ok or i % 20 == 0:
        print(f"pose {i}: R={R} ok={ok} digests {d_b} {d_o} {d_s}", flush=True)
print("poses", poses, "mismatching frames", bad)
sys.exit(1 if bad else 0)
