"""Per chunk of 1024 depth-consecutive visible Gaussians (coarse_emit_kernel's unit of work): entries (blocks touched) per chunk
and the biggest Gaussian's blocks — which chunks could take a from-the-Gaussian's-side placement."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer

scene_name = sys.argv[1] if len(sys.argv) > 1 else "garden_like"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5_834_784
W, H = (int(v) for v in sys.argv[4].split("x")) if len(sys.argv) > 4 else (1920, 1080)
sc, near, far, pos, label = bench.make_scene(scene_name, n, torch.device("cuda:0"))
if len(sys.argv) > 3 and sys.argv[3] != "-":
    pos = tuple(float(v) for v in sys.argv[3].split(","))
cam = camera.default_camera(W, H, near=near, far=far, position=pos)
r = SplatRasterizer(W, H)
r.configure_from_scene(sc)
r.draw(cam, plan="blocks")
g = r.map_geometry_state()
gx, gy = (W + 15) // 16, (H + 15) // 16
m = g["means2D"]; ext = r.rects.to(torch.float32); vis = g["tilesTouched"] > 0
x0 = ((m[:, 0] - ext[:, 0]) / 16).to(torch.int32).clamp(0, gx); x1 = ((m[:, 0] + ext[:, 0] + 15) / 16).to(torch.int32).clamp(0, gx)
y0 = ((m[:, 1] - ext[:, 1]) / 16).to(torch.int32).clamp(0, gy); y1 = ((m[:, 1] + ext[:, 1] + 15) / 16).to(torch.int32).clamp(0, gy)
depth = g["depths"][vis]
order = torch.argsort(depth.view(torch.int32), stable=True)
x0, x1, y0, y1 = (v[vis][order] for v in (x0, x1, y0, y1))
blocks = (((x1 - 1) // 8 - x0 // 8 + 1) * ((y1 - 1) // 8 - y0 // 8 + 1)).to(torch.int64)
V = blocks.numel()
pad = (-V) % 1024
b = torch.cat([blocks, torch.zeros(pad, dtype=torch.int64, device=blocks.device)]).view(-1, 1024)
tot, mx = b.sum(1), b.max(1).values
E = int(tot.sum())
print(label, pos, f"{W}x{H}", "V", V, "E", E, "chunks", b.shape[0])
for cap in (4096, 6144, 8192, 12288):
    for lb in (8, 16, 32, 64, 1 << 20):
        ok = (tot <= cap) & (mx <= lb)
        print(f"  cap {cap:6d} max blocks {lb:8d}: chunks {float(ok.float().mean()):.3f}  entries {float(tot[ok].sum()) / E:.3f}")
q = torch.tensor([0.1, 0.5, 0.9, 0.99], device=tot.device)
print("  entries per chunk quantiles", torch.quantile(tot.float(), q).tolist(), "max blocks per chunk quantiles", torch.quantile(mx.float(), q).tolist())
big = (b > 16)
print("  Gaussians with more than 16 blocks per chunk: mean", float(big.sum(1).float().mean()), "their share of the entries", float((b * big).sum()) / E)
