"""Two frames in flight: two host threads, a rasterizer and a stream each, render the headline frame over and over — what a viewer
that overlaps frame i's emission (bound by the HBM write path) with frame i + 1's per-Gaussian stage (latency-bound) gets.
python scripts/two_in_flight.py [frames per thread]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W, H = 1920, 1080
sc, near, far, pos, label = bench.make_scene("garden_like", 5_834_784, torch.device("cuda:0"))
cam = camera.default_camera(W, H, near=near, far=far, position=pos)
rasts = [SplatRasterizer(W, H), SplatRasterizer(W, H)]
for r in rasts:
    r.configure_from_scene(sc)
    for _ in range(20):
        r.draw(cam)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    rasts[0].draw(cam)
one = (time.perf_counter() - t0) / frames * 1e3
ref = rasts[0].out_color.clone()
go = threading.Barrier(3)
def work(t):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(10):
            rasts[t].draw(cam)
        go.wait()
        for _ in range(frames):
            rasts[t].draw(cam)
        s.synchronize()
    go.wait()
ths = [threading.Thread(target=work, args=(t,)) for t in range(2)]
for th in ths: th.start()
go.wait(); t0 = time.perf_counter(); go.wait(); two = (time.perf_counter() - t0) / (2 * frames) * 1e3
for th in ths: th.join()
torch.cuda.synchronize()
print(f"one frame at a time {one:.4f} ms a frame ({1e3 / one:.0f} fps); two in flight {two:.4f} ms a frame ({1e3 / two:.0f} fps); "
      f"pixels equal: {bool(torch.equal(rasts[0].out_color, ref)) and bool(torch.equal(rasts[1].out_color, ref))}")
