"""What one rank of a sharded frame does: per-band frame time and stage times (single GPU, one band at a time), for uniform
bands and for bands re-cut by instance count (what the ranks converge to, sharding.balanced_bands).
usage: band_timings.py [width height]"""
import sys, os  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, time, bench
from gsrast_amd import camera, sharding
from gsrast_amd.rasterizer import SplatRasterizer
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
gx, gy = (W + 15) // 16, (H + 15) // 16
sc, near, far, pos, label = bench.make_scene("garden_like", 5_834_784, torch.device("cuda", 0))
cam = camera.default_camera(W, H, near=near, far=far, position=pos)
r = SplatRasterizer(W, H); r.configure_from_scene(sc)
r.draw(cam)
rg = r.map_image_state()["ranges"].to(torch.int64)
row_cost = (rg[:, 1] - rg[:, 0]).clamp(min=0).view(gy, gx).sum(1).cpu().numpy().astype(np.float64)
proj = {"uniform_ms": {}, "recut_ms": {}}
print("#", label, f"{W}x{H}; {gy} tile rows; times in ms; 'uniform' = even split of the tile rows, 'recut' = split by instances per tile row")
for world in (1, 2, 4, 8):
    for kind in (("uniform",) if world == 1 else ("uniform", "recut")):
        b = sharding.uniform_bands(gy, world) if kind == "uniform" else sharding.balanced_bands(row_cost, world, floor_cost=0.02 * row_cost.mean() + 1.0)
        worst = 0.0
        for g in range(world):
            rows = (b[g], b[g + 1])
            for _ in range(3): r.draw(cam, tile_rows=rows)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(15): r.draw(cam, tile_rows=rows)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 15 * 1e3
            r.draw(cam, tile_rows=rows, profile=True)
            worst = max(worst, dt)
            print(f"ranks={world} {kind} band={rows} R={r.last_num_rendered} plan={r.last_plan} frame={dt:.3f}", {k: round(v, 3) for k, v in r.last_stage_ms.items() if v})
        print(f"ranks={world} {kind}: slowest band {worst:.3f} ms (+ one exchange of the row bands per frame)")
        proj[kind + "_ms"][str(world)] = round(worst, 4)
        if world == 1:
            proj["recut_ms"]["1"] = round(worst, 4)
# what bench.py reports as config.projected_ms beside a measured N-GPU number (profiles/band_projection.json, keyed by resolution)
import json
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "band_projection.json")
allp = json.load(open(path)) if os.path.exists(path) else {}
allp[f"{W}x{H}"] = proj
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(allp, open(path, "w"), indent=1)
