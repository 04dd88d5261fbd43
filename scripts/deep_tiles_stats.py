"""What four waves do to a slow tile (csrc/blend.hip, deep tiles): per-tile times of the blend with one wave per tile and with
the history's deep tiles, same pose, same box. python scripts/deep_tiles_stats.py [z] (GSR_DEEP in the environment: the rule)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer
dev = torch.device("cuda:0")
sc, near, far, pos, label = bench.make_scene("garden_like", bench.DEFAULT_SPLATS, dev)
r = SplatRasterizer(1920, 1080, device=dev)
r.configure_from_scene(sc)
for z in [float(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["-30"])]:
    cam = camera.default_camera(1920, 1080, near=near, far=far, position=(0.0, 0.0, z))
    def run(**kw):
        for _ in range(10):
            r.draw(cam, **kw)
        ms = 0.0
        for _ in range(10):
            r.draw(cam, profile=True, **kw)
            ms += r.last_stage_ms["blend"] / 10
        t, d, n = r.tile_history_times()
        return ms, t.astype(np.float64) / 100.0, d, n
    ms0, t0, d0, n0 = run(deep_tiles=False)
    ms1, t1, d1, n1 = run()
    print(f"eye (0,0,{z:g})  GSR_DEEP={os.environ.get('GSR_DEEP')}: blend {ms0:.4f} -> {ms1:.4f} ms; deep tiles {int(d1.sum())} (order says {n1}) of {t0.size}")
    print(f"  one wave per tile: mean {t0.mean():.1f} us, quantiles 50/90/99/max {np.percentile(t0, 50):.0f}/{np.percentile(t0, 90):.0f}/{np.percentile(t0, 99):.0f}/{t0.max():.0f}, sum/5120 {t0.sum() / 5120:.0f} us")
    if d1.any():
        ratio = t0[d1] / np.maximum(t1[d1], 1e-3)
        print(f"  the deep tiles: one wave {np.median(t0[d1]):.0f} us (median) -> four waves {np.median(t1[d1]):.0f} us; ratio quantiles 10/50/90: "
              f"{np.percentile(ratio, 10):.2f}/{np.percentile(ratio, 50):.2f}/{np.percentile(ratio, 90):.2f}; longest deep {t1[d1].max():.0f} us, longest ordinary {t1[~d1].max():.0f} us")
        top = np.argsort(-t0)[:10]
        print("  ten longest tiles (one wave -> now, deep?):", ", ".join(f"{t0[i]:.0f}->{t1[i]:.0f}{'D' if d1[i] else ''}" for i in top))
