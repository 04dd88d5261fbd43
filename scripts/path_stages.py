"""Stage times along bench.py's camera path (every 4th pose, profiled frames behind two unprofiled ones of the same pose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gsrast_amd.rasterizer import SplatRasterizer
W, H = 1920, 1080
sc, near, far, pos, label = bench.make_scene("garden_like", 5_834_784, torch.device("cuda:0"))
r = SplatRasterizer(W, H)
r.configure_from_scene(sc)
cams = bench.walk_path(W, H, near, far)
print("# pose  R/V   R          plan feed ovl | frame  | preprocess scan depth_order duplicate masks/pass2 ranges blend | frame - sum")
import time
for i in range(0, len(cams), 4):
    for _ in range(3):
        r.draw(cams[i])
    t0 = time.perf_counter(); r.draw(cams[i]); ms = (time.perf_counter() - t0) * 1e3
    r.draw(cams[i], profile=True)
    st = r.last_stage_ms
    V = int((r.map_geometry_state()["tilesTouched"] != 0).sum().item())
    keys = ("preprocess", "scan", "depth_order", "duplicate", "sort_pass1", "sort_pass2", "ranges", "blend")
    vals = [st.get(k, 0.0) for k in keys]
    hidden = r.last_emit_overlapped
    tot = sum(vals) - (min(st.get("blend", 0.0), st.get("duplicate", 0.0)) if hidden else 0.0)
    print(f"{i:5d} {r.last_num_rendered / max(V, 1):6.1f} {r.last_num_rendered:10d}  {r.last_plan:6s} {'lists' if r.last_blend_from_lists else 'blocks':6s} {int(hidden)} | {ms:6.3f} | "
          + " ".join(f"{v:6.3f}" for v in vals) + f" | {ms - tot:6.3f}")
