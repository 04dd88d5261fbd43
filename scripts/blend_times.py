"""Blend stage times (HIP events, mean of 10 frames) of the bench scene from the poses the bench reports, with and without
faint splats. python scripts/blend_times.py   (GSR_LIB_TAG selects a tagged build)"""
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
print("library", os.path.basename(_capi.LIB_PATH))
def t(title, cam, **kw):
    for _ in range(3):
        r.draw(cam, **kw)
    ms, tot = 0.0, 0.0
    for _ in range(10):
        r.draw(cam, profile=True, **kw)
        ms += r.last_stage_ms["blend"] / 10
        tot += sum(r.last_stage_ms.values()) / 10
    st = r.map_image_state()
    digest = (int(r.out_color.view(torch.int32).to(torch.int64).sum()), int(st["nContrib"].to(torch.int64).sum()), int(st["finalT"].view(torch.int32).to(torch.int64).sum()))
    print(f"  blend {ms:.4f} ms  (all stages {tot:.3f})  {title}  plan={r.last_plan}{'+lists' if r.last_blend_from_lists else ''} digest={digest}")
cams = {p: camera.default_camera(1920, 1080, near=near, far=far, position=(0.0, 0.0, p)) for p in (-5.0, -14.0, -30.0, -50.0)}
for p, c in cams.items():
    t(f"eye (0,0,{p:g})", c)
r.opacities = r.opacities * 0.1
t("eye (0,0,-5), opacities x 0.1", cams[-5.0])
t("eye (0,0,-14), opacities x 0.1", cams[-14.0])
