"""What one rank of a sharded 1080p frame does: per-band frame time and stage times (single GPU, one band at a time)."""
import sys, os  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time, bench
from gsrast_amd import camera, sharding
from gsrast_amd.rasterizer import SplatRasterizer
sc, near, far, pos, label = bench.make_scene("garden_like", 5_834_784, torch.device("cuda", 0))
cam = camera.default_camera(1920, 1080, near=near, far=far, position=pos)
r = SplatRasterizer(1920, 1080); r.configure_from_scene(sc)
print("#", label, "1920x1080; bands = uniform split of the 68 tile rows; times in ms")
for world in (1, 2, 4, 8):
    b = sharding.uniform_bands(68, world)
    worst = 0.0
    for g in range(world):
        rows = (b[g], b[g + 1])
        for _ in range(3): r.draw(cam, tile_rows=rows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15): r.draw(cam, tile_rows=rows)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 15 * 1e3
        r.draw(cam, tile_rows=rows, profile=True)
        worst = max(worst, dt)
        print(f"ranks={world} band={rows} R={r.last_num_rendered} frame={dt:.3f}", {k: round(v, 3) for k, v in r.last_stage_ms.items() if v})
    print(f"ranks={world}: slowest band {worst:.3f} ms (+ one all-gather of the row bands per frame)")
