"""Blend stage with and without deep tiles (four waves per slow tile, one walk: csrc/blend.hip), HIP-event times over 10
frames after the history has settled, bench scene from the poses whose frames end on a few slow tiles.
python scripts/deep_ab.py [GSR_DEEP="gain_x16,frac_x16,floor_ticks" in the environment for another rule]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gsrast_amd import camera, _capi
from gsrast_amd.rasterizer import SplatRasterizer
dev = torch.device("cuda:0")
sc, near, far, pos, label = bench.make_scene("garden_like", bench.DEFAULT_SPLATS, dev)
r = SplatRasterizer(1920, 1080, device=dev)
r.configure_from_scene(sc)
print("library", os.path.basename(_capi.LIB_PATH), "GSR_DEEP =", os.environ.get("GSR_DEEP"))
def t(title, cam, **kw):
    for _ in range(8):
        r.draw(cam, **kw)
    ms, tot = 0.0, 0.0
    for _ in range(10):
        r.draw(cam, profile=True, **kw)
        ms += r.last_stage_ms["blend"] / 10
        tot += sum(r.last_stage_ms.values()) / 10
    st = r.map_image_state()
    digest = (int(r.out_color.view(torch.int32).to(torch.int64).sum()), int(st["nContrib"].to(torch.int64).sum()), int(st["finalT"].view(torch.int32).to(torch.int64).sum()))
    hs = r.tile_history_stats() if hasattr(r, "tile_history_stats") else None
    print(f"  blend {ms:.4f} ms  (all stages {tot:.3f})  {title}  plan={r.last_plan}{'+lists' if r.last_blend_from_lists else ''} deep={r.last_deep_tiles} reordered={r.last_tiles_reordered} digest={digest} hist={hs}")
poses = [float(p) for p in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("-14", "-20", "-30", "-50"))]
for p in poses:
    cam = camera.default_camera(1920, 1080, near=near, far=far, position=(0.0, 0.0, p))
    t(f"eye (0,0,{p:g}) deep off", cam, deep_tiles=False)
    t(f"eye (0,0,{p:g}) deep auto", cam)
    t(f"eye (0,0,{p:g}) deep all", cam, deep_tiles="all")
if "--faint" in sys.argv:
    r.opacities = r.opacities * 0.1
    for p in (-5.0, -14.0):
        cam = camera.default_camera(1920, 1080, near=near, far=far, position=(0.0, 0.0, p))
        t(f"eye (0,0,{p:g}) opacities x 0.1, deep off", cam, deep_tiles=False)
        t(f"eye (0,0,{p:g}) opacities x 0.1, deep auto", cam)
