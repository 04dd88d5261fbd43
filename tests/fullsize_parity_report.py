"""One-off full-size parity run (BASELINE config 2 stand-in, 5.83 M splats, 1920x1080): the HIP
frame against the CPU oracle on the same inputs. Takes a few minutes of host time (the oracle's
stable sort of R keys is single-threaded), so it is a script, not a pytest case.
Usage (GPU box): python tests/fullsize_parity_report.py > gpurun_out/fullsize_parity.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer
from oracle import cpu_oracle

scene = scenes.garden_like_scene(5_834_784, 43)
span = float(np.max(scene["means3D"][:, :3].max(0) - scene["means3D"][:, :3].min(0)))
cam = camera.default_camera(1920, 1080, near=0.001 * span, far=span)
t0 = time.time()
tm = {}
exp = cpu_oracle.forward(scene, cam, (0.0, 0.0, 0.0), threads=min(64, cpu_oracle.hardware_concurrency()), timings=tm)
print(f"oracle: R={exp['num_rendered']} R_f={exp['records_staged']} in {time.time() - t0:.1f}s {tm}")
r = SplatRasterizer(1920, 1080)
r.configure_from_scene(scene)
img = r.draw(cam, count_staged=True).cpu().numpy()
print(f"hip   : R={r.last_num_rendered} R_f={r.last_records_staged}")
g = {k: v.cpu().numpy() for k, v in r.map_geometry_state().items()}
for k in ("radii", "tilesTouched", "pointOffsets"):
    print(f"  {k:14s} bit-exact: {np.array_equal(g[k].view(exp[k].dtype), exp[k])}")
for k in ("means2D", "depths", "cov3D", "rgb", "conicOpacity"):
    print(f"  {k:14s} bit-exact: {np.array_equal(g[k], exp[k])}")
b = r.map_binning_state()
print("  sorted keys    bit-exact:", np.array_equal(b["keys"].cpu().numpy().view(np.uint64), exp["keys"]))
print("  sorted values  bit-exact:", np.array_equal(b["values"].cpu().numpy().view(np.uint32), exp["values"]))
im = {k: v.cpu().numpy() for k, v in r.map_image_state().items()}
print("  tile ranges    bit-exact:", np.array_equal(im["ranges"].view(np.uint32), exp["ranges"]))
d = np.abs(img.astype(np.float64) - exp["out_color"])
per_pixel = d.max(axis=0)
print(f"  image max abs err {d.max():.3e}; pixels over 1e-4: {(per_pixel > 1e-4).sum()} of {per_pixel.size}; "
      f"99.99th pct {np.percentile(per_pixel, 99.99):.3e}")
print(f"  finalT max abs err {np.abs(im['finalT'] - exp['finalT']).max():.3e}; nContrib mismatches "
      f"{(im['nContrib'].view(np.uint32) != exp['nContrib']).sum()} of {per_pixel.size}")
