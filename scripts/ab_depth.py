"""A/B of the two depth orders (GSR_FLAG_DEPTH_LSD vs the two-pass bucket order) on the bench scene: same lists bit for
bit, stage times and frame times side by side. python scripts/ab_depth.py [--splats N] [--width W --height H] [--pose x,y,z]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gsrast_amd import camera
from gsrast_amd.rasterizer import SplatRasterizer

p = argparse.ArgumentParser()
p.add_argument("--splats", type=int, default=bench.DEFAULT_SPLATS)
p.add_argument("--width", type=int, default=1920)
p.add_argument("--height", type=int, default=1080)
p.add_argument("--scene", default="garden_like")
p.add_argument("--poses", default="0,0,-5;0,0,-14;0,0,-30")
p.add_argument("--steps", type=int, default=30)
p.add_argument("--rows", default="", help="tile rows b0,b1 (a band)")
args = p.parse_args()
dev = torch.device("cuda:0")
sc, near, far, pos, label = bench.make_scene(args.scene, args.splats, dev)
r = SplatRasterizer(args.width, args.height, device=dev)
r.configure_from_scene(sc)
rows = tuple(int(v) for v in args.rows.split(",")) if args.rows else None
for pose in args.poses.split(";"):
    eye = tuple(float(v) for v in pose.split(","))
    cam = camera.look_from(eye, args.width, args.height, near=near, far=far) if hasattr(camera, "look_from") else camera.default_camera(args.width, args.height, near=near, far=far, position=eye)
    res = {}
    for mode in ("lsd", "auto"):
        for _ in range(3):
            r.draw(cam, depth_order=mode, tile_rows=rows)
        b = r.map_binning_state()
        keys, vals = b["keys"].clone(), b["values"].clone()
        img = r.out_color.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r.draw(cam, depth_order=mode, tile_rows=rows)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        st = {}
        for _ in range(10):
            r.draw(cam, depth_order=mode, profile=True, tile_rows=rows)
            for k, v in r.last_stage_ms.items():
                st[k] = st.get(k, 0.0) + v / 10
        res[mode] = (keys, vals, img, ms, st, r.last_depth_buckets, r.last_num_rendered, r.last_plan)
    a, b = res["lsd"], res["auto"]
    same = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]))
    print(f"pose {pose} R={b[6]} plan={b[7]} buckets_used={b[5]} identical={same}")
    for mode in ("lsd", "auto"):
        ms, st = res[mode][3], res[mode][4]
        print(f"  {mode:5s} {ms:.3f} ms/frame  " + " ".join(f"{k}={v:.3f}" for k, v in st.items() if v > 0))
    r.draw(cam, depth_order="auto", tile_rows=rows)
    bs = r.depth_bucket_stats()
    c = bs["counts"]
    print(f"  buckets={bs['buckets']} samples={bs['samples']} entries={int(c.sum())} mean={c.mean():.0f} max={int(c.max())} "
          f"over4096={int((c > 4096).sum())} p99={int(np.percentile(c, 99))} empty={int((c == 0).sum())}")
    assert same and b[5] and not a[5]
