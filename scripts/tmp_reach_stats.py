import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer
dev = torch.device("cuda:0")
def stats(title, sc, near, far, poses):
    r = SplatRasterizer(1920, 1080, device=dev)
    r.configure_from_scene(sc)
    for z in poses:
        cam = camera.default_camera(1920, 1080, near=near, far=far, position=(0.0, 0.0, z))
        r.draw(cam)
        g = r.map_geometry_state()
        vis = g["tilesTouched"] != 0
        co = g["conicOpacity"][vis].double(); xy = g["means2D"][vis].double()
        A, B, C, op = co[:, 0], co[:, 1], co[:, 2], co[:, 3]
        thr = torch.log(255.0 * op).clamp(min=0.0) + 2e-3
        det = (A * C - B * B).clamp(min=1e-12)
        rx, ry = torch.sqrt(2 * thr * C / det), torch.sqrt(2 * thr * A / det)
        nx = torch.floor(xy[:, 0] + rx) - torch.ceil(xy[:, 0] - rx) + 1
        ny = torch.floor(xy[:, 1] + ry) - torch.ceil(xy[:, 1] - ry) + 1
        V = int(vis.sum())
        out = [f"{title} z={z}: V={V} R/V={r.last_num_rendered / V:.1f} feed={'lists' if r.last_blend_from_lists else 'blocks'}"]
        for lim in (1, 2, 3, 4, 6, 8):
            out.append(f"bbox<={lim}x{lim}: {float(((nx <= lim) & (ny <= lim)).double().mean()):.3f}")
        out.append(f"cols<=4: {float((nx <= 4).double().mean()):.3f} cols<=8: {float((nx <= 8).double().mean()):.3f}")
        print("  ".join(out), flush=True)
sc, near, far, pos, label = bench.make_scene("garden_like", bench.DEFAULT_SPLATS, dev)
stats("garden", sc, near, far, (-14.0, -20.0, -30.0, -40.0))
del sc
tl = scenes.trained_like(bench.DEFAULT_SPLATS, seed=45)
span = float(np.max(tl["means3D"][:, :3].max(0) - tl["means3D"][:, :3].min(0)))
stats("trained_like", tl, 0.001 * span, span, (-10.0, -20.0, -48.0))
