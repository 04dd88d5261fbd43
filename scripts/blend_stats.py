"""Per-tile work of the blend stage on the bench frame: list lengths, last contributors, batches staged."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer
W, H = 1920, 1080
sc, near, far, pos, label = bench.make_scene("garden_like", 5_834_784)
cam = camera.default_camera(W, H, near=near, far=far, position=pos)
r = SplatRasterizer(W, H)
r.configure_from_scene(sc)
r.draw(cam, count_staged=True)
im = r.map_image_state()
rg = im["ranges"].to(torch.int64)
length = (rg[:, 1] - rg[:, 0]).view(68, 120)
nc = im["nContrib"].to(torch.int64)
pad = torch.zeros(68 * 16, 120 * 16, dtype=torch.int64, device=nc.device); pad[:1080, :1920] = nc
tile_last = pad.view(68, 16, 120, 16).permute(0, 2, 1, 3).reshape(68, 120, 256).max(-1).values
ft = im["finalT"]; padt = torch.zeros(68 * 16, 120 * 16, device=ft.device); padt[:1080, :1920] = ft
tile_maxT = padt.view(68, 16, 120, 16).permute(0, 2, 1, 3).reshape(68, 120, 256).max(-1).values
q = torch.tensor([0.5, 0.9, 0.99, 1.0], device=nc.device)
print("list length quantiles", torch.quantile(length.float().flatten(), q).tolist())
print("tile last-contributor quantiles", torch.quantile(tile_last.float().flatten(), q).tolist(), "sum", int(tile_last.sum()))
print("records staged (R_f)", r.last_records_staged)
# tiles that never finish (some pixel still above the cut-off at the end of the list) walk their whole list
unfinished = (tile_maxT >= 0.001)
print("tiles walking their whole list:", int(unfinished.sum()), "their list lengths:", torch.quantile(length[unfinished].float(), q).tolist())
print("walked records per tile (approx) quantiles:", torch.quantile(torch.where(unfinished, length, tile_last).float().flatten(), q).tolist())
