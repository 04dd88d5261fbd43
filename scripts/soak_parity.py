"""Parity soak: random first-person poses of a mid-size scene, every frame against the CPU oracle (oracle/gsr_oracle.cpp):
sorted lists, ranges, transmittance and nContrib bit for bit, pixels within 2e-6 — the claim of DESIGN.md §5 outside the
BASELINE poses. Test infrastructure (imports oracle/).  python scripts/soak_parity.py [poses] [splats] [width height] [scene]
scene: garden_like (default) | trained_like (scenes.trained_like: flat splats on surfaces, heavy-tailed scales, huge background splats)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer
from oracle import cpu_oracle

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (960, 540)
which = sys.argv[5] if len(sys.argv) > 5 else "garden_like"
scene = scenes.trained_like(n, seed=77) if which == "trained_like" else scenes.garden_like_scene(n, seed=77)
span = float(np.max(scene["means3D"][:, :3].max(0) - scene["means3D"][:, :3].min(0)))
r = SplatRasterizer(W, H, background=(0.2, 0.1, 0.3))
r.configure_from_scene(scene)
rng = np.random.default_rng(12)
threads = max(1, min(64, cpu_oracle.hardware_concurrency()))
worst_px, frames, t_words, flips, list_bad, deep_frames = 0.0, 0, 0, 0, 0, 0
for i in range(poses):
    pos = tuple(float(x) for x in rng.uniform(-7, 7, 3))
    if i % 3 == 2:
        pos = tuple(4.0 * x for x in pos)                     # every third pose from far out: a tile or two per splat — deep tiles (csrc/blend.hip)
    scale = float(rng.choice([1.0, 1.0, 0.3, 0.1]))          # every fourth frame or so with faint splats: deep walks
    # looking at the cloud's centre, give or take half a radian (front = (cos p sin y, sin p, cos p cos y), FirstPersonCamera.cpp:30)
    d = -np.asarray(pos) / max(np.linalg.norm(pos), 1e-6)
    yaw, pitch = float(np.arctan2(d[0], d[2]) + rng.uniform(-0.5, 0.5)), float(np.arcsin(np.clip(d[1], -1, 1)) + rng.uniform(-0.3, 0.3))
    cam = camera.first_person_camera(pos, yaw, pitch, float(np.radians(45.0)), 0.001 * span, span, W, H, True)
    sc = dict(scene, opacities=(scene["opacities"] * np.float32(scale)).astype(np.float32))
    exp = cpu_oracle.forward(sc, cam, (0.2, 0.1, 0.3), threads=threads)
    if exp["num_rendered"] == 0:
        continue
    r.opacities = __import__("torch").from_numpy(sc["opacities"]).to(r.device)
    img = r.draw(cam, count_staged=True).cpu().numpy()
    frames += 1
    deep_frames += int(getattr(r, "last_deep_tiles", False))
    b, im = r.map_binning_state(), r.map_image_state()
    ok_lists = (r.last_num_rendered == exp["num_rendered"] and r.last_records_staged == exp["records_staged"]
                and np.array_equal(b["keys"].cpu().numpy().view(np.uint64), exp["keys"])
                and np.array_equal(b["values"].cpu().numpy().view(np.uint32), exp["values"])
                and np.array_equal(im["ranges"].cpu().numpy().view(np.uint32), exp["ranges"]))
    list_bad += 0 if ok_lists else 1
    t_words += int((im["finalT"].cpu().numpy().view(np.uint32) != exp["finalT"].view(np.uint32)).sum())
    flips += int((im["nContrib"].cpu().numpy().view(np.uint32) != exp["nContrib"]).sum())
    worst_px = max(worst_px, float(np.abs(img - exp["out_color"]).max()))
    if i % 10 == 0:
        print(f"pose {i}: R={exp['num_rendered']} R_f={exp['records_staged']} plan={r.last_plan} opacity x {scale} lists_ok={ok_lists} worst pixel so far {worst_px:.2e}", flush=True)
print(f"{frames} frames with R > 0 ({which}, {n} splats, {W}x{H}; {deep_frames} of them composited by four waves per tile): frames whose lists / ranges / R / R_f differ: {list_bad}; finalT words differing: {t_words}; "
      f"nContrib flips: {flips}; largest pixel difference: {worst_px:.3e}")
sys.exit(0 if (list_bad == 0 and t_words == 0 and flips == 0 and worst_px <= 2e-6) else 1)
