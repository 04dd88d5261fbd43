"""Shared helpers for the parity tests."""
import os

import numpy as np

from gsrast_amd.camera import Camera

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

INT_KEYS = ["radii", "tilesTouched", "pointOffsets", "rects", "keys_unsorted", "values_unsorted", "keys", "values",
            "ranges", "nContrib"]
FLT_KEYS = ["means2D", "depths", "cov3D", "rgb", "conicOpacity", "finalT", "out_color"]


def load_golden(name="config1.npz"):
    z = np.load(os.path.join(ROOT, "tests", "golden", name))
    n = z["in_means3D"].shape[0]
    shs = np.zeros((n, 48), np.float32)
    shs[:, :3] = z["in_shs_dc"]
    scene = {"means3D": z["in_means3D"], "scales": z["in_scales"], "rotations": z["in_rotations"],
             "opacities": z["in_opacities"], "shs": shs}
    cam = Camera(view=z["cam_view"], proj=z["cam_proj"], cam_pos=z["cam_pos"], tan_fovx=float(z["cam_tan"][0]),
                 tan_fovy=float(z["cam_tan"][1]), width=int(z["size"][0]), height=int(z["size"][1]))
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    exp["num_rendered"] = int(z["num_rendered"])
    exp["records_staged"] = int(z["records_staged"])
    return scene, cam, tuple(float(v) for v in z["background"]), exp


def single_gaussian_scene(pos=(0.0, 0.0, 0.0), scale=0.1, opacity=0.8, dc=(1.0, 0.5, -0.5), quat=(1, 0, 0, 0), n=1):
    """n copies of one isotropic Gaussian (positions may be overridden by the caller)."""
    means = np.ones((n, 4), np.float32)
    means[:, :3] = np.asarray(pos, np.float32)
    scales = np.full((n, 4), np.e, np.float32)
    scales[:, :3] = scale
    rot = np.tile(np.asarray(quat, np.float32), (n, 1))
    shs = np.zeros((n, 48), np.float32)
    shs[:, :3] = np.asarray(dc, np.float32)
    return {"means3D": means, "scales": scales, "rotations": rot,
            "opacities": np.full(n, opacity, np.float32), "shs": shs}


def image_report(got, exp, tol=1e-4):
    """Max abs error plus the pixels over tolerance (threshold-flip candidates)."""
    d = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    per_pixel = d.max(axis=0) if d.ndim == 3 else d
    bad = per_pixel > tol
    return float(d.max()), int(bad.sum()), per_pixel
