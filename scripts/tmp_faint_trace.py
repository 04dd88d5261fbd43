import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer
dev = torch.device("cuda:0")
sc, near, far, pos, label = bench.make_scene("garden_like", bench.DEFAULT_SPLATS, dev)
r = SplatRasterizer(1920, 1080, device=dev)
r.configure_from_scene(sc)
cam = camera.default_camera(1920, 1080, near=near, far=far, position=(0.0, 0.0, -5.0))
for _ in range(4):
    r.draw(cam)
r.opacities = r.opacities * 0.1
for i in range(16):
    r.draw(cam, profile=True)
    print(i, f"blend {r.last_stage_ms['blend']:.3f} lists={r.last_blend_from_lists} ovl={r.last_emit_overlapped} hist={r.tile_history_stats()}")
