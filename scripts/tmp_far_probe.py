import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer
dev = torch.device("cuda:0")
def run(r, cam, **kw):
    for _ in range(8):
        r.draw(cam, **kw)
    ms = 0.0
    for _ in range(8):
        r.draw(cam, profile=True, **kw)
        ms += r.last_stage_ms["blend"] / 8
    return ms
def probe(title, sc, near, far, cams):
    r = SplatRasterizer(1920, 1080, device=dev)
    r.configure_from_scene(sc)
    for name, cam in cams:
        res = {m: run(r, cam, deep_tiles=m) for m in (False, "all", "all8", "all16", None)}
        hs = r.tile_history_stats()
        V = max(1, int((r.map_geometry_state()["radii"] > 0).sum()))
        print(f"{title} {name}: R/V {r.last_num_rendered / V:.1f} blend one wave {res[False]:.4f}  4 waves {res['all']:.4f}  8 {res['all8']:.4f}  16 {res['all16']:.4f}  auto {res[None]:.4f}"
              f"  [tiles x mean / longest = {8160 * hs['mean_ticks'] / max(1, hs['longest_ticks']):.0f}]", flush=True)
tl = scenes.trained_like(bench.DEFAULT_SPLATS, seed=45)
span = float(np.max(tl["means3D"][:, :3].max(0) - tl["means3D"][:, :3].min(0)))
nf = (0.001 * span, span)
rng = np.random.default_rng(3)
cams = [(f"z={z}", camera.default_camera(1920, 1080, near=nf[0], far=nf[1], position=(0.0, 0.0, z))) for z in (-20.0, -32.0, -48.0, -70.0)]
probe("trained_like", tl, nf[0], nf[1], cams)
del tl
sc, near, far, pos, label = bench.make_scene("garden_like", bench.DEFAULT_SPLATS, dev)
cams = [(f"z={z}", camera.default_camera(1920, 1080, near=near, far=far, position=(0.0, 0.0, z))) for z in (-25.0, -30.0, -40.0, -50.0, -70.0)]
probe("garden", sc, near, far, cams)
