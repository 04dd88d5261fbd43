"""Deep tiles (csrc/blend.hip): the blend with one wave per tile against four (eight, sixteen) waves that share a tile's walk,
HIP-event times of the blend stage over 8 frames behind 8 warm-up frames of the same pose (the view's tile history settled),
same box for every figure. 1920 x 1080; the bench scene walking in from far away, then far views of the trained-like scene.
python scripts/deep_tiles_table.py   (profiles/r06_deep_tiles.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer
dev = torch.device("cuda:0")
W, H = 1920, 1080


def run(r, cam, **kw):
    for _ in range(8):
        r.draw(cam, **kw)
    ms, frame = 0.0, 0.0
    for _ in range(8):
        r.draw(cam, profile=True, **kw)
        ms += r.last_stage_ms["blend"] / 8
    return ms


def table(title, sc, near, far, zs):
    r = SplatRasterizer(W, H, device=dev)
    r.configure_from_scene(sc)
    print(f"# {title}\n# eye (0,0,z)   R/V    feed             one wave   4 waves   8 waves  16 waves |  library's own   [tiles x mean / longest of its history]")
    for z in zs:
        cam = camera.default_camera(W, H, near=near, far=far, position=(0.0, 0.0, z))
        res = {m: run(r, cam, deep_tiles=m) for m in (False, "all", "all8", "all16")}
        auto = run(r, cam)
        feed = r.last_plan + ("+lists" if r.last_blend_from_lists else "") + ("+overlap" if r.last_emit_overlapped else "") + ("+deep" if r.last_deep_tiles else "")
        hs = r.tile_history_stats()
        V = max(1, int((r.map_geometry_state()["radii"] > 0).sum()))
        if r.last_num_rendered == 0:
            continue
        same_feed = r.last_blend_from_lists or r.last_plan == "sort"
        print(f"  {z:7.1f}   {r.last_num_rendered / V:6.1f}  {feed:22s} {res[False]:8.4f}  {res['all']:8.4f}  {res['all8']:8.4f}  {res['all16']:8.4f} | {auto:8.4f}"
              f"        [{8160 * hs['mean_ticks'] // max(1, hs['longest_ticks'])}]" + ("" if same_feed else "   (block-fed: no deep tiles, the flags change nothing)"), flush=True)


sc, near, far, pos, label = bench.make_scene("garden_like", bench.DEFAULT_SPLATS, dev)
table("bench scene (garden-like stand-in, 5 834 784 splats)", sc, near, far, (-50.0, -40.0, -30.0, -25.0, -22.0, -20.0, -17.0, -14.0, -12.0))
del sc
tl = scenes.trained_like(bench.DEFAULT_SPLATS, seed=45)
span = float(np.max(tl["means3D"][:, :3].max(0) - tl["means3D"][:, :3].min(0)))
table("trained-like scene (scenes.trained_like, 5 834 784 splats: flat splats on surfaces, 500 huge background splats)", tl, 0.001 * span, span,
      (-70.0, -48.0, -40.0, -32.0, -20.0, -10.0))
