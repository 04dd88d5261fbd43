"""Lane utilisation of a one-lane-per-Gaussian loop over its tile blocks, in depth order, on the bench scene (two poses)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer
W, H = 1920, 1080
dev = torch.device("cuda", 0)
sc, near, far, pos, label = bench.make_scene("garden_like", 5_834_784, dev)
for pose in [pos, (0.0, 0.0, -14.0)]:
    cam = camera.default_camera(W, H, near=near, far=far, position=pose)
    r = SplatRasterizer(W, H)
    r.configure_from_scene(sc)
    r.draw(cam, plan="blocks")
    g = r.map_geometry_state()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    m = g["means2D"]; ext = r.rects.to(torch.float32); vis = g["tilesTouched"] > 0
    x0 = ((m[:, 0] - ext[:, 0]) / 16).to(torch.int32).clamp(0, gx); x1 = ((m[:, 0] + ext[:, 0] + 15) / 16).to(torch.int32).clamp(0, gx)
    y0 = ((m[:, 1] - ext[:, 1]) / 16).to(torch.int32).clamp(0, gy); y1 = ((m[:, 1] + ext[:, 1] + 15) / 16).to(torch.int32).clamp(0, gy)
    d = g["depths"][vis]
    x0, x1, y0, y1 = x0[vis], x1[vis], y0[vis], y1[vis]
    blocks = (((x1 - 1) // 8 - x0 // 8 + 1) * ((y1 - 1) // 8 - y0 // 8 + 1)).to(torch.int64)
    order = torch.argsort(d, stable=True)
    b = blocks[order]
    n = b.numel() // 64 * 64
    bw = b[:n].view(-1, 64)
    mx = bw.max(1).values; sm = bw.sum(1)
    print("pose", pose, "V", b.numel(), "E", int(b.sum()), "waves", bw.shape[0], "sum of wave max", int(mx.sum()), "lane utilisation", float(sm.sum()) / float(mx.sum() * 64))
    q = torch.tensor([0.5, 0.9, 0.99], device=b.device)
    print(" blocks/gaussian quantiles", torch.quantile(b.float()[::5], q).tolist(), "max", int(b.max()), " wave-max quantiles", torch.quantile(mx.float(), q).tolist())
    # per 1024-chunk: sum of wave max (serial iterations per CU slot)
    c = mx[: mx.numel() // 16 * 16].view(-1, 16).sum(1)
    print(" per-chunk sum of wave max: mean", float(c.float().mean()), "max", int(c.max()), "first 5", c[:5].tolist())
