import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer
W, H = 1920, 1080
sc, near, far, pos, label = bench.make_scene("garden_like", 5_834_784)
cam = camera.default_camera(W, H, near=near, far=far, position=pos)
r = SplatRasterizer(W, H); r.configure_from_scene(sc); r.draw(cam)
g = r.map_geometry_state()
gx, gy = 120, 68
m = g["means2D"]; ext = r.rects.to(torch.float32); vis = g["tilesTouched"] > 0
x0 = ((m[:, 0] - ext[:, 0]) / 16).to(torch.int32).clamp(0, gx); x1 = ((m[:, 0] + ext[:, 0] + 15) / 16).to(torch.int32).clamp(0, gx)
y0 = ((m[:, 1] - ext[:, 1]) / 16).to(torch.int32).clamp(0, gy); y1 = ((m[:, 1] + ext[:, 1] + 15) / 16).to(torch.int32).clamp(0, gy)
x0, x1, y0, y1 = x0[vis].long(), x1[vis].long(), y0[vis].long(), y1[vis].long()
R = int(((x1 - x0) * (y1 - y0)).sum())
for bw, bh in ((8, 8), (4, 16), (2, 32), (16, 4), (8, 16), (4, 32), (16, 8)):
    E = int((((x1 - 1) // bw - x0 // bw + 1) * ((y1 - 1) // bh - y0 // bh + 1)).sum())
    nb = ((gx + bw - 1) // bw) * ((gy + bh - 1) // bh)
    print(f"block {bw}x{bh}: blocks {nb}, E {E/1e6:.2f} M, keys/entry {R/E:.1f}, density {R/E/(bw*bh):.2f}")
