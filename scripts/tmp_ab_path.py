"""A/B along the camera path: blend + frame, every 4th pose, library under GSR_LIB_TAG vs current; run twice with different tags."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gsrast_amd.rasterizer import SplatRasterizer
from gsrast_amd import _capi
W, H = 1920, 1080
sc, near, far, pos, label = bench.make_scene("garden_like", 5_834_784, torch.device("cuda:0"))
r = SplatRasterizer(W, H)
r.configure_from_scene(sc)
cams = bench.walk_path(W, H, near, far)
kw = {}
if os.environ.get("AB_DEEP") == "off":
    kw["deep_tiles"] = False
print("# lib", os.path.basename(_capi.LIB_PATH), kw)
rows = []
for i in range(0, 72, 4):
    for _ in range(4):
        r.draw(cams[i], **kw)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); r.draw(cams[i], **kw); ts.append((time.perf_counter() - t0) * 1e3)
    b = 0.0
    for _ in range(4):
        r.draw(cams[i], profile=True, **kw); b += r.last_stage_ms["blend"] / 4
    print(f"{i:4d} frame {np.median(ts):.4f} blend {b:.4f} {'lists' if r.last_blend_from_lists else 'blocks'} ovl={int(r.last_emit_overlapped)}", flush=True)
