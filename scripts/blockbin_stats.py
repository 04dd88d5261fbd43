"""Block-binning workload figures for the bench frame: entries E, units, keys per entry, per-block spread."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer

# usage: blockbin_stats.py [scene [splats [x,y,z [WxH]]]]
scene_name = sys.argv[1] if len(sys.argv) > 1 else "garden_like"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5_834_784
W, H = (int(v) for v in sys.argv[4].split("x")) if len(sys.argv) > 4 else (1920, 1080)
sc, near, far, pos, label = bench.make_scene(scene_name, n, torch.device("cuda:0"))
if len(sys.argv) > 3 and sys.argv[3] != "-":
    pos = tuple(float(v) for v in sys.argv[3].split(","))
cam = camera.default_camera(W, H, near=near, far=far, position=pos)
r = SplatRasterizer(W, H)
r.configure_from_scene(sc)
r.draw(cam, plan="blocks")
g = r.map_geometry_state()
gx, gy = (W + 15) // 16, (H + 15) // 16
m = g["means2D"]; ext = r.rects.to(torch.float32); vis = g["tilesTouched"] > 0
x0 = ((m[:, 0] - ext[:, 0]) / 16).to(torch.int32).clamp(0, gx); x1 = ((m[:, 0] + ext[:, 0] + 15) / 16).to(torch.int32).clamp(0, gx)
y0 = ((m[:, 1] - ext[:, 1]) / 16).to(torch.int32).clamp(0, gy); y1 = ((m[:, 1] + ext[:, 1] + 15) / 16).to(torch.int32).clamp(0, gy)
x0, x1, y0, y1 = x0[vis], x1[vis], y0[vis], y1[vis]
tiles = ((x1 - x0) * (y1 - y0)).to(torch.int64)
blocks = (((x1 - 1) // 8 - x0 // 8 + 1) * ((y1 - 1) // 8 - y0 // 8 + 1)).to(torch.int64)
print(label, pos, f"{W}x{H}", "V", int(vis.sum()), "R", int(tiles.sum()), r.last_num_rendered, "E", int(blocks.sum()), "units(2048)", int(blocks.sum()) // 2048)
print("keys/entry", float(tiles.sum()) / float(blocks.sum()), "w mean", float((x1 - x0).float().mean()), "h mean", float((y1 - y0).float().mean()))
q = torch.tensor([0.5, 0.9, 0.99, 0.999], device=tiles.device)
print("tiles quantiles", torch.quantile(tiles.float()[::7], q).tolist(), "max", int(tiles.max()))
rg = r.map_image_state()["ranges"].to(torch.int64)
per_tile = (rg[:, 1] - rg[:, 0]).view(gy, gx)
pb = torch.zeros((gy + 7) // 8, (gx + 7) // 8, dtype=torch.int64, device=per_tile.device)
for by in range(pb.shape[0]):
    for bx in range(pb.shape[1]):
        pb[by, bx] = per_tile[by * 8:(by + 1) * 8, bx * 8:(bx + 1) * 8].sum()
print("keys per block: min", int(pb.min()), "mean", float(pb.float().mean()), "max", int(pb.max()))
