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
    return ms
def probe(title, sc, near, far, poses):
    r = SplatRasterizer(1920, 1080, device=dev)
    r.configure_from_scene(sc)
    for z in poses:
        cam = camera.default_camera(1920, 1080, near=near, far=far, position=(0.0, 0.0, z))
        a = run(r, cam, deep_tiles=False)
        feed = r.last_plan + ("+lists" if r.last_blend_from_lists else "") + ("+ovl" if r.last_emit_overlapped else "")
        b = run(r, cam, deep_tiles="all")
        feed_b = r.last_plan + ("+lists" if r.last_blend_from_lists else "") + ("+ovl" if r.last_emit_overlapped else "")
        c = run(r, cam, deep_tiles="all", tile_history=False)
        d = run(r, cam, deep_tiles=False, tile_history=False)
        e = run(r, cam)
        print(f"{title} z={z}: blend ordinary {a:.4f} [{feed}] (no history {d:.4f}), all deep {b:.4f} [{feed_b}] (no history {c:.4f}), auto {e:.4f}; R/V {r.last_num_rendered / max(1, int((r.map_geometry_state()['radii'] > 0).sum())):.1f}", flush=True)
sc, near, far, pos, label = bench.make_scene("garden_like", bench.DEFAULT_SPLATS, dev)
probe("garden", sc, near, far, (-9.0, -12.0, -14.0, -17.0, -20.0, -25.0, -30.0, -40.0))
del sc
tl = scenes.trained_like(bench.DEFAULT_SPLATS, seed=45)
span = float(np.max(tl["means3D"][:, :3].max(0) - tl["means3D"][:, :3].min(0)))
probe("trained_like", tl, 0.001 * span, span, (-3.0, -6.0, -10.0, -20.0))
