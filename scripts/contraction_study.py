"""FMA-contraction sensitivity of the forward path.

The reference binary is built by nvcc with its default -fmad=true (apps/gsrast/gscuda/CMakeLists.txt:1-13 sets no
-fmad=false); this repo's oracle and kernels are built with -ffp-contract=off. The study runs the oracle source
built both ways (oracle/Makefile: -ffp-contract=off vs -ffp-contract=fast -mfma) on the same inputs and counts what
moves: radii, rects, tilesTouched, instances, list entries, pixels over 1e-4. It is the one bound available on the
distance between "bit-exact against the oracle" and the real reference binary.

  python scripts/contraction_study.py          prints the table (CPU only)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np


def compare(scene, cam, bg=(0.0, 0.0, 0.0), threads=8):
    from oracle import cpu_oracle
    a = cpu_oracle.forward(scene, cam, bg, threads=threads)
    b = cpu_oracle.forward(scene, cam, bg, threads=threads, contract=True)
    n = scene["means3D"].shape[0]
    vis = (a["tilesTouched"] > 0) | (b["tilesTouched"] > 0)
    rep = {
        "splats": n, "visible": int(vis.sum()), "instances": int(a["num_rendered"]), "instances_contract": int(b["num_rendered"]),
        "radii_differ": int((a["radii"] != b["radii"]).sum()),
        "rects_differ": int((a["rects"] != b["rects"]).any(axis=1)[vis].sum()),
        "tiles_touched_differ": int((a["tilesTouched"] != b["tilesTouched"]).sum()),
        "visibility_flips": int(((a["tilesTouched"] > 0) != (b["tilesTouched"] > 0)).sum()),
    }
    for k in ("means2D", "depths", "cov3D", "conicOpacity", "rgb"):
        x, y = a[k][vis], b[k][vis]
        rep[f"{k}_values_differ"] = int((x != y).sum())
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.abs(x.astype(np.float64) - y) / np.maximum(np.abs(x.astype(np.float64)), 1e-30)
        rep[f"{k}_max_rel"] = float(np.nanmax(rel)) if rel.size else 0.0
    # list entries that moved: (tile, Gaussian) pairs present on one side only, and positions whose value differs
    pa = (a["keys"] >> np.uint64(32)).astype(np.uint64) << np.uint64(32) | a["values"].astype(np.uint64)
    pb = (b["keys"] >> np.uint64(32)).astype(np.uint64) << np.uint64(32) | b["values"].astype(np.uint64)
    rep["pairs_only_one_side"] = int(np.setxor1d(pa, pb).size)
    m = min(pa.size, pb.size)
    rep["list_positions_differ"] = int((a["values"][:m] != b["values"][:m]).sum()) + abs(pa.size - pb.size)
    # order flips among pairs both sides have: per tile, is the relative order of common Gaussians the same?
    d = np.abs(a["out_color"].astype(np.float64) - b["out_color"]).max(axis=0)
    rep["pixels_over_1e-4"] = int((d > 1e-4).sum())
    rep["pixels_over_1e-6"] = int((d > 1e-6).sum())
    rep["max_pixel_diff"] = float(d.max())
    rep["ncontrib_differ"] = int((a["nContrib"] != b["nContrib"]).sum())
    rep["records_staged"] = (int(a["records_staged"]), int(b["records_staged"]))
    return rep


def frames():
    from gsrast_amd import camera, scenes
    yield "config 1 (1000 isotropic, 128x128)", scenes.isotropic_scene(1000, 42), camera.default_camera(128, 128), (0.0, 0.0, 0.0)
    sc = scenes.garden_like_scene(150_000, seed=43)
    sc["means3D"][:, 2] += 6.0
    yield ("150k garden-like, 1920x1080 (the frame of test_1080p_midsize_frame_against_oracle)", sc,
           camera.default_camera(1920, 1080, near=0.05, far=80.0), (0.05, 0.05, 0.05))


if __name__ == "__main__":
    for name, sc, cam, bg in frames():
        rep = compare(sc, cam, bg)
        print(name)
        for k, v in rep.items():
            print(f"  {k:28s} {v}")
