import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer
dev = torch.device("cuda:0")
def run(r, cam, **kw):
    for _ in range(8):
        r.draw(cam, **kw)
    ms, tot = 0.0, 0.0
    for _ in range(10):
        r.draw(cam, profile=True, **kw)
        ms += r.last_stage_ms["blend"] / 10
        tot += sum(r.last_stage_ms.values()) / 10
    return ms, tot
sc, near, far, pos, label = bench.make_scene("garden_like", bench.DEFAULT_SPLATS, dev)
print("GSR_BLOCK_FEED_MIN", os.environ.get("GSR_BLOCK_FEED_MIN"), "GSR_DEEP_ALL_MAX", os.environ.get("GSR_DEEP_ALL_MAX"))
for (w, h) in ((1920, 1080), (3840, 2160)):
    r = SplatRasterizer(w, h, device=dev)
    r.configure_from_scene(sc)
    for z in (-5.0, -7.0, -9.0, -12.0):
        cam = camera.default_camera(w, h, near=near, far=far, position=(0.0, 0.0, z))
        a, ta = run(r, cam, overlap_emit=False)
        fa = r.last_plan + ("+lists" if r.last_blend_from_lists else "")
        b, tb = run(r, cam)
        fb = r.last_plan + ("+lists" if r.last_blend_from_lists else "") + ("+ovl" if r.last_emit_overlapped else "")
        print(f"{w}x{h} z={z}: serial blend {a:.4f} all {ta:.3f} [{fa}]   default blend {b:.4f} all {tb:.3f} [{fb}]", flush=True)
    del r
