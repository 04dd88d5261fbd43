"""Per-rank cost of a tile-row band on one GPU (what each rank of an N-way sharded frame pays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gsrast_amd import camera, scenes, sharding
from gsrast_amd.rasterizer import SplatRasterizer
scene = scenes.garden_like_scene(5_834_784, 43)
span = float(np.max(scene["means3D"][:, :3].max(0) - scene["means3D"][:, :3].min(0)))
cam = camera.default_camera(1920, 1080, near=0.001 * span, far=span)
r = SplatRasterizer(1920, 1080)
r.configure_from_scene(scene)
r.draw(cam)
rg = r.map_image_state()["ranges"].to(torch.int64)
per_row = (rg[:, 1] - rg[:, 0]).view(68, 120).sum(1).cpu().numpy().astype(np.float64)
for world in (1, 2, 4, 8):
    for name, bounds in (("uniform", sharding.uniform_bands(68, world)), ("balanced", sharding.balanced_bands(per_row, world, 1.0))):
        worst = 0.0
        for g in range(world):
            rows = (bounds[g], bounds[g + 1])
            for _ in range(3): r.draw(cam, tile_rows=rows)
            t0 = time.perf_counter()
            for _ in range(10): r.draw(cam, tile_rows=rows)
            ms = (time.perf_counter() - t0) / 10 * 1e3
            worst = max(worst, ms)
        print(f"world={world} {name:8s} bounds={bounds} slowest rank {worst:.3f} ms")
