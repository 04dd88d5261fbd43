"""Soak of what the library does by default on its second stream — the blend beside the emission, geomState.rgb beside the
depth sort or the blend, the slow tiles of the frame before first — against the same call with every kernel on the caller's
stream in patch order (GSR_FLAG_SERIAL_EMIT | GSR_FLAG_NO_TILE_HISTORY): a camera that walks, jumps and looks away, sizes
that alternate; image, finalT, nContrib, geomState.rgb, numRendered and the sorted lists bit for bit.
python scripts/soak_defaults.py [frames] [splats] [garden_like | trained_like]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
which = sys.argv[3] if len(sys.argv) > 3 else "garden_like"
scene = scenes.trained_like(n, seed=91) if which == "trained_like" else scenes.garden_like_scene(n, seed=91)
inner = np.abs(scene["means3D"][:, :3]).max(1) < 20.0          # (not the trained-like scene's background shell)
span = float(np.max(scene["means3D"][inner, :3].max(0) - scene["means3D"][inner, :3].min(0)))
big = dict(scene, scales=scene["scales"].copy())
big["scales"][:, :3] *= 5.0                               # dense frames: 48 and more instances per visible Gaussian, the block-fed blend
rs = []
for sc in (scene, big):
    for w, h in ((960, 540), (1280, 720)):
        r = SplatRasterizer(w, h, background=(0.2, 0.1, 0.3))
        r.configure_from_scene(sc)
        r.base_opacities = sc["opacities"]
        rs.append(r)
rng = np.random.default_rng(5)
pos = np.array([0.0, 0.0, -6.0])
bad = counts = 0
seen = {"tiles_reordered": 0, "emit_overlapped": 0, "colors_beside": 0, "R0": 0}


def state(r):
    st = r.map_image_state()
    out = [r.out_color.clone(), st["finalT"].clone(), st["nContrib"].clone(), r.map_geometry_state()["rgb"].clone()]
    if r.last_num_rendered:
        b = r.map_binning_state()
        out += [b["keys"].clone(), b["values"].clone()]
    return out


for i in range(frames):
    if i % 37 == 36:
        pos = rng.uniform(-12, 12, 3)                     # a jump
    else:
        pos = pos + rng.normal(0, 0.15, 3)                # a walk
    away = i % 53 == 52                                   # looks away: R == 0
    d = (pos if away else -pos) / max(np.linalg.norm(pos), 1e-6)
    yaw, pitch = float(np.arctan2(d[0], d[2])), float(np.arcsin(np.clip(d[1], -1, 1)))
    r = rs[(i // 11) % 4]
    cam = camera.first_person_camera(tuple(float(x) for x in pos), yaw, pitch, float(np.radians(45.0)), 0.001 * span, 4.0 * span, r.width, r.height, True)
    scale = np.float32(0.1 if i % 17 == 16 else 1.0)
    r.opacities = torch.from_numpy((r.base_opacities * scale).astype(np.float32)).to(r.device)
    plan = "blocks" if i % 5 == 4 else "auto"
    r.draw(cam, plan=plan, overlap_emit=False, tile_history=False)
    ref, R = state(r), r.last_num_rendered
    for t in state(r)[:4]:
        t.fill_(0)                                        # (poison what the second call must write again)
    r.out_color.fill_(float("nan")); r.map_geometry_state()["rgb"].fill_(float("nan"))
    r.draw(cam, plan=plan)
    got = state(r)
    counts += 1
    for k in ("tiles_reordered", "emit_overlapped", "colors_beside"):
        seen[k] += int(getattr(r, "last_" + k))
    seen["R0"] += int(R == 0)
    same = r.last_num_rendered == R and len(got) == len(ref) and all(
        torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)
        for a, b in zip(got, ref) if R or a.shape != ref[0].shape)     # (R == 0 leaves out_color untouched: GSCuda.cu:775-778)
    if R == 0:
        same = same and bool((got[3] == 0).all())
    bad += int(not same)
    if i % 40 == 0:
        print(f"frame {i}: {r.width}x{r.height} R={R} plan={r.last_plan} reordered={r.last_tiles_reordered} overlapped={r.last_emit_overlapped} same={same}", flush=True)
print(f"{which}: {counts} frames, {bad} differing; second-stream work seen: {seen}")
