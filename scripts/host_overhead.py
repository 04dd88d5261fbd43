"""Host cost of one SplatRasterizer.draw() call: a frame so small that the device is never the bottleneck
(1 000 splats, 128 x 128), timed over many calls — what the Python / ctypes layer and the library's own host code add to
every frame (it matters once a sharded frame is down to 0.6 ms). Prints microseconds per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer

r = SplatRasterizer(128, 128)
r.configure_from_scene(scenes.isotropic_scene(1000, 42))
cam = camera.default_camera(128, 128)
for sync in (True, False):
    for _ in range(50):
        r.draw(cam, sync=sync)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n):
        r.draw(cam, sync=sync)
    torch.cuda.synchronize()
    print(f"draw(sync={sync}): {(time.perf_counter() - t0) / n * 1e6:.1f} us per call (tiny frame: host-bound)")
r.draw(cam, profile=True)
print("device stages of that frame (ms):", {k: round(v, 4) for k, v in r.last_stage_ms.items() if v})
