"""Per-call host times of the headline frame (ms; o = blend beside the emission, r = tiles reordered): what do the first calls of a fresh
rasterizer take, the first after 200 ms of idling, and those of a second rasterizer on a chip that is busy? The tile history has
settled after two calls; the dozen slower frames after an idle gap are the chip's clocks coming back (bench.py --spinup)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer
W, H = 1920, 1080
sc, near, far, pos, label = bench.make_scene("garden_like", 5_834_784, torch.device("cuda:0"))
cam = camera.default_camera(W, H, near=near, far=far, position=pos)
def run(r, n, tag):
    ts, fl = [], []
    for i in range(n):
        t0 = time.perf_counter(); r.draw(cam); ts.append((time.perf_counter() - t0) * 1e3)
        fl.append(("o" if r.last_emit_overlapped else "-") + ("r" if r.last_tiles_reordered else "-"))
    print(tag, " ".join("%.3f%s" % (t, f) for t, f in zip(ts, fl)))
r = SplatRasterizer(W, H); r.configure_from_scene(sc)
run(r, 32, "fresh rasterizer, first calls :")
run(r, 16, "straight on                   :")
time.sleep(0.2)
run(r, 24, "after 200 ms of idle          :")
r2 = SplatRasterizer(W, H); r2.configure_from_scene(sc)
run(r, 8, "old one again                 :")
run(r2, 24, "second rasterizer (busy chip) :")
