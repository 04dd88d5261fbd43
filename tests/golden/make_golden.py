"""Mints tests/golden/config1.npz: BASELINE config 1 (1000 isotropic Gaussians, 128x128,
reference default camera) inputs plus every intermediate and the image.

Expected values come from the numpy restatement (oracle/oracle_np.py); the script refuses to
write the file unless the independent C++ oracle agrees (integers exactly, floats to 1e-6).
Neither is the reference itself — 42yeah/GSRast has no tests or vectors for this path and
cannot be built here — so the fixture pins this repo's two restatements against each other,
not against reference output. Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gsrast_amd import camera, scenes          # noqa: E402
from oracle import cpu_oracle, oracle_np       # noqa: E402

INT_KEYS = ["radii", "tilesTouched", "pointOffsets", "rects", "keys_unsorted", "values_unsorted", "keys", "values",
            "ranges", "nContrib"]
FLT_KEYS = ["means2D", "depths", "cov3D", "rgb", "conicOpacity", "finalT", "out_color"]


def main():
    scene = scenes.isotropic_scene(1000, 42)
    cam = camera.default_camera(128, 128, near=0.01, far=100.0)
    bg = (0.0, 0.0, 0.0)
    a = oracle_np.forward(scene, cam, bg)
    b = cpu_oracle.forward(scene, cam, bg)
    for k in INT_KEYS:
        assert np.array_equal(a[k], b[k]), k
    for k in FLT_KEYS:
        assert np.abs(a[k].astype(np.float64) - b[k]).max() <= 1e-6, k
    assert a["num_rendered"] == b["num_rendered"] and a["records_staged"] == b["records_staged"]
    out = {"in_" + k: v for k, v in scene.items() if k != "shs"}
    out["in_shs_dc"] = scene["shs"][:, :3].copy()      # the other 45 floats per splat are zero
    out.update(cam_view=cam.view, cam_proj=cam.proj, cam_pos=cam.cam_pos,
               cam_tan=np.array([cam.tan_fovx, cam.tan_fovy], np.float32),
               size=np.array([cam.width, cam.height], np.int32), background=np.array(bg, np.float32),
               num_rendered=np.array(a["num_rendered"], np.int64),
               records_staged=np.array(a["records_staged"], np.int64))
    for k in INT_KEYS + FLT_KEYS:
        out["exp_" + k] = a[k]
    path = os.path.join(ROOT, "tests", "golden", "config1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; R =", a["num_rendered"], "R_f =", a["records_staged"])


if __name__ == "__main__":
    main()
