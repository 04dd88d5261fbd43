"""Are the library's switch points the right ones on a scene that looks like a trained one? ("heuristic surface".)
For every pose of a random first-person tour of scenes.trained_like (or garden_like): the frame's instances per visible
Gaussian (R / V), what the library chose by itself — binning plan, the blend's feed, blend beside the emission — and the
frame time under every forced alternative:
    sort            GSR_FLAG_PLAN_SORT
    blocks          GSR_FLAG_PLAN_BLOCKS | GSR_FLAG_SERIAL_EMIT   (blend fed from the sorted lists below 48 instances per Gaussian, else from the block lists)
    blocks+overlap  GSR_FLAG_PLAN_BLOCKS | GSR_FLAG_OVERLAP_EMIT  (blend fed from the block lists, beside the emission)
    auto            no flag
(every variant with this rasterizer's tile history, warmed up on the pose: slow tiles first under all of them)
and whether "auto" is within 5 % of the best of them. Host wall time per frame incl. the device synchronise, median of 5 after 4 (a camera cut is over for the tile history after three frames).
Usage: python scripts/thresholds_check.py [scene] [splats] [poses] [width height]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer

which = sys.argv[1] if len(sys.argv) > 1 else "trained_like"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
poses = int(sys.argv[3]) if len(sys.argv) > 3 else 60
W, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080)
scene = scenes.trained_like(n, seed=45) if which == "trained_like" else scenes.garden_like_scene(n, seed=43)
fg = np.abs(scene["means3D"][:, :3]).max(1) < 20.0                     # (not the background shell)
span = float(np.max(scene["means3D"][fg, :3].max(0) - scene["means3D"][fg, :3].min(0)))
near, far = 0.001 * span, 4.0 * span
r = SplatRasterizer(W, H)
r.configure_from_scene(scene)
rng = np.random.default_rng(21)
variants = {"sort": dict(plan="sort"), "blocks": dict(plan="blocks", overlap_emit=False),
            "blocks+overlap": dict(plan="blocks", overlap_emit=True), "auto": dict()}


def timed(cam, kw):
    for _ in range(4):
        r.draw(cam, **kw)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r.draw(cam, **kw)
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def plans_agree(cam):
    """sorted lists, ranges, T, nContrib and pixels of the two binning plans, bit for bit (scripts/stress_plans_check.py's check, per pose)"""
    out = []
    for plan in ("sort", "blocks"):
        img = r.draw(cam, plan=plan).clone()
        b, im = r.map_binning_state(), r.map_image_state()
        out.append((img.view(torch.int32), b["keys"].clone(), b["values"].clone(), im["ranges"].clone(), im["finalT"].view(torch.int32).clone(), im["nContrib"].clone()))
    return all(torch.equal(x, y) for x, y in zip(*out))


print(f"# {which}, N={n}, {W}x{H}, {poses} poses; times in ms")
print("# pose  R/V     R          auto: plan feed overlapped |   sort  blocks  blocks+overlap   auto | best      auto/best")
within, rows, agree, checked = 0, 0, 0, 0
for i in range(poses):
    eye = rng.normal(size=3) * np.array([3.0, 0.6, 3.0])
    if i % 5 == 4:
        eye *= 3.0                                                       # (every fifth pose from further out,
    if i % 7 == 6:
        eye *= 4.0 / max(np.linalg.norm(eye) / 12.0, 1e-3) if np.linalg.norm(eye) < 48.0 else 1.0   # every seventh from 48 units away: few tiles per splat)
    d = -eye / max(np.linalg.norm(eye), 1e-6)
    yaw, pitch = float(np.arctan2(d[0], d[2]) + rng.uniform(-0.6, 0.6)), float(np.arcsin(np.clip(d[1], -1, 1)) + rng.uniform(-0.3, 0.3))
    cam = camera.first_person_camera(tuple(eye), yaw, pitch, float(np.radians(45.0)), near, far, W, H, True)
    t = {k: timed(cam, kw) for k, kw in variants.items()}
    if r.last_num_rendered == 0:
        continue
    auto_state = (r.last_plan, r.last_blend_from_lists, r.last_emit_overlapped)
    stages = {}
    for k, kw in variants.items():
        if k != "auto":
            r.draw(cam, profile=True, **kw)
            stages[k] = dict(r.last_stage_ms)
    V = int((r.map_geometry_state()["tilesTouched"] != 0).sum().item())
    best = min(t, key=lambda k: t[k] if k != "auto" else 1e9)
    ratio = t["auto"] / t[best]
    rows += 1
    within += int(ratio <= 1.05)
    if i % 4 == 0:                                   # (every fourth pose: the 267 M-pair frames take a while to compare)
        checked += 1
        agree += int(plans_agree(cam))
    print(f"{i:5d} {r.last_num_rendered / max(V, 1):6.1f} {r.last_num_rendered:10d}  {auto_state[0]:6s} {'lists ' if auto_state[1] else 'blocks'} {int(auto_state[2])}"
          f"          | {t['sort']:6.3f} {t['blocks']:6.3f} {t['blocks+overlap']:6.3f}          {t['auto']:6.3f} | {best:14s} {ratio:5.3f}"
          f" | serial blocks: emit {stages['blocks'].get('duplicate', 0):.3f} blend {stages['blocks'].get('blend', 0):.3f}", flush=True)
    if ratio > 1.05 and os.environ.get("GSR_SHOW_STAGES"):
        for k, st in stages.items():
            print(f"#        {k:14s} " + " ".join(f"{n}={v:.3f}" for n, v in st.items() if v > 0), flush=True)
print(f"# auto within 5 % of the best forced variant on {within} of {rows} poses ({100.0 * within / max(rows, 1):.0f} %)")
print(f"# sort plan and block plan bit-equal (sorted keys / values, ranges, finalT, nContrib, pixels) on {agree} of {checked} poses checked")
