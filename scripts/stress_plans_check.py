import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer
W, H = 1920, 1080
sc, near, far, pos, label = bench.make_scene("stress", 50_000_000)
cam = camera.default_camera(W, H, near=near, far=far, position=pos)
r = SplatRasterizer(W, H)
r.configure_from_scene(sc)
img_b = r.draw(cam, plan="blocks").clone()
b = r.map_binning_state()
kb, vb = b["keys"].clone(), b["values"].clone()
rg = r.map_image_state()["ranges"].clone()
img_s = r.draw(cam, plan="sort")
b = r.map_binning_state()
print(label, "R", r.last_num_rendered, "keys equal", bool(torch.equal(b["keys"], kb)), "values equal", bool(torch.equal(b["values"], vb)),
      "ranges equal", bool(torch.equal(r.map_image_state()["ranges"], rg)), "pixels equal", bool(torch.equal(img_s, img_b)))
g = r.backward(torch.ones((3, H, W)))
print("backward finite:", all(bool(torch.isfinite(v).all()) for v in g.values()), "dL_dcolors sum", float(g["dL_dcolors"].sum()))
