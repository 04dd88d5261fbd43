import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer
z, mode = float(sys.argv[1]), sys.argv[2]
dev = torch.device("cuda:0")
sc, near, far, pos, label = bench.make_scene("garden_like", bench.DEFAULT_SPLATS, dev)
r = SplatRasterizer(1920, 1080, device=dev)
r.configure_from_scene(sc)
cam = camera.default_camera(1920, 1080, near=near, far=far, position=(0.0, 0.0, z))
kw = {"deep_tiles": {"all": "all", "off": False, "auto": None}[mode]}
for _ in range(12):
    r.draw(cam, **kw)
print("done", z, mode, r.last_plan, r.last_blend_from_lists, r.last_deep_tiles)
