"""How alike are the tile times of consecutive frames? (tile_order_kernel, csrc/blend.hip: the order is dropped below 0.5.)
Renders the bench scene along bench.py's camera path and through its random views, every frame once, and prints what the
rasterizer's tile history found per frame: similarity of the two last frames (sum of minima over sum of maxima of the
tiles' times), whether the order is dropped, the plan. Usage: python scripts/history_similarity.py [splats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from gsrast_amd import scenes
from gsrast_amd.rasterizer import SplatRasterizer

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.DEFAULT_SPLATS
W, H = 1920, 1080
sc = scenes.garden_like_scene(n, seed=43)
span = float(np.max(sc["means3D"][:, :3].max(0) - sc["means3D"][:, :3].min(0)))
near, far = 0.001 * span, span
r = SplatRasterizer(W, H)
r.configure_from_scene(sc)
for name, cams in (("path", bench.walk_path(W, H, near, far)), ("random_views", bench.random_views(W, H, near, far)),
                   ("path, every frame twice", [c for c in bench.walk_path(W, H, near, far) for _ in (0, 1)])):
    sims, dropped = [], 0
    print(f"## {name}: {len(cams)} frames")
    for i, cam in enumerate(cams):
        r.draw(cam)
        torch.cuda.synchronize()
        st = r.tile_history_stats()
        sims.append(st["similarity"])
        dropped += int(r.last_tile_order_dropped)
        if i % 6 == 0 or name == "random_views":
            print(f"frame {i:3d} plan={r.last_plan:6s} lists_feed={int(r.last_blend_from_lists)} overlapped={int(r.last_emit_overlapped)} reordered={int(r.last_tiles_reordered)} "
                  f"dropped={int(r.last_tile_order_dropped)} similarity(last sort)={st['similarity']:.3f} mean_tile_us={st['mean_ticks'] / 100:.1f} R={r.last_num_rendered}")
    s = np.asarray(sims)
    print(f"# {name}: similarity min {s.min():.3f} p10 {np.percentile(s, 10):.3f} median {np.median(s):.3f} max {s.max():.3f}; frames with the order dropped: {dropped}")
