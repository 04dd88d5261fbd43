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


PIXEL_TOL = 2e-6     # measured 3.6e-7: the colour sums are three fused multiply-adds on alpha T where the reference rounds twice


def assert_blend_parity(img, final_t, n_contrib, exp, what="", bitwise_t=True):
    """The tile loop's outputs against the oracle's. The HIP blend evaluates power, the exponential (as glibc's expf: the
    oracle's), alpha, T and every test in the reference's float32 operation order, so the transmittance and the last
    contributor are the oracle's BIT FOR BIT — no pixel sits on the other side of a threshold — and the colours differ by the
    rounding of the fused sums only: far inside the north star's 1e-4. bitwise_t=False: `exp` comes from the committed fixture,
    minted by the numpy restatement (whose float32 exp is not libm's): transmittance within 1e-6 then."""
    nc = np.asarray(n_contrib).view(np.uint32)
    flips = int((nc != exp["nContrib"]).sum())
    ft = np.asarray(final_t)
    t_diff = int((ft.view(np.uint32) != exp["finalT"].view(np.uint32)).sum())
    max_err, n_bad, _ = image_report(np.asarray(img), exp["out_color"], 1e-4)
    print(f"[parity] {what}: max abs pixel err {max_err:.3e}, pixels > 1e-4: {n_bad}, finalT words differing: {t_diff}, nContrib flips: {flips}")
    assert flips == 0 and (t_diff == 0 if bitwise_t else float(np.abs(ft - exp["finalT"]).max()) <= 1e-6), (what, flips, t_diff)
    assert n_bad == 0 and max_err <= PIXEL_TOL, (what, max_err, n_bad)
    return max_err


def check_backward_chain(got, g, scene, cam, w, h, ids, upstream=None, tol_scale=1.0):
    """gsr_backward's per-Gaussian chain (cov2D -> cov3D -> scales / rotations, pixel centre and Jacobian -> means3D)
    for the Gaussians `ids`, against oracle/backward_np.py fed with the GPU's own upstream gradients — or, with `upstream`
    (dict: dL_dmean2D, dL_dconic_opacity, dL_dcov2D indexable like `got`), with sums computed independently of the GPU: an
    end-to-end check. `got`: the gradient arrays (numpy, full size or indexable by id), `g`: geometry state arrays (cov3D),
    `scene`: host inputs. Returns the largest expected magnitude of (dL_dcov3D, dL_dmeans3D, dL_dscales, dL_drotations)."""
    from oracle import backward_np as B
    focal = h / (2.0 * cam.tan_fovy)
    m = len(ids)
    exp_cov, exp_mean, exp_scale, exp_rot = np.zeros((m, 6)), np.zeros((m, 3)), np.zeros((m, 3)), np.zeros((m, 4))
    for j, i in enumerate(ids):
        c3 = g["cov3D"][i].astype(np.float64)
        m3 = scene["means3D"][i, :3].astype(np.float64)
        up = got if upstream is None else upstream
        dconic = up["dL_dconic_opacity"][i, :3].astype(np.float64)
        # (the chain starts from the summed dL/dcov2D where the call produced it: gsr_backward_args.dL_dcov2D)
        dcov = up["dL_dcov2D"][i].astype(np.float64) if "dL_dcov2D" in up else None
        exp_cov[j] = B.conic_backward(c3, m3, cam.view, focal, cam.tan_fovx, cam.tan_fovy, dconic, dcov)
        exp_mean[j] = (B.project_mean2d_backward(m3, cam.proj, w, h, up["dL_dmean2D"][i].astype(np.float64)) +
                       B.conic_backward_mean(c3, m3, cam.view, focal, cam.tan_fovx, cam.tan_fovy, dconic, dcov))
        exp_scale[j], exp_rot[j] = B.cov3d_backward(scene["scales"][i, :3], scene["rotations"][i], 1.0,
                                                    (got["dL_dcov3D"][i] if upstream is None else exp_cov[j]).astype(np.float64))
    ids = np.asarray(ids)
    # compared per Gaussian relative to its own magnitude (float32 outputs of a chain evaluated in double). End to end
    # (`upstream`): the rotation gradient of a nearly round splat is a difference of nearly equal products — it vanishes for a
    # round one — so that what is left of the sums' sixth digit shows in its third: ten times the tolerance of the others.
    for name, e, gotv, tol in (("dL_dcov3D", exp_cov, got["dL_dcov3D"][ids], 2e-3),
                               ("dL_dmeans3D", exp_mean, got["dL_dmeans3D"][ids][:, :3], 3e-3),
                               ("dL_dscales", exp_scale, got["dL_dscales"][ids][:, :3], 3e-3),
                               ("dL_drotations", exp_rot, got["dL_drotations"][ids], 3e-3 if upstream is None else 3e-2)):
        if m == 0:
            continue
        err = np.abs(gotv - e).max(1)
        mag = np.maximum(np.abs(e).max(1), 1e-3 * np.abs(e).max())
        if upstream is not None:
            print(f"[backward] end to end, {name}: worst error {float((err / np.maximum(mag, 1e-30)).max()):.2e} of the Gaussian's own largest component")
        assert (err <= tol * tol_scale * np.maximum(mag, 1e-30)).all(), (name, float((err / np.maximum(mag, 1e-30)).max()))
    return [float(np.abs(e).max()) if m else 0.0 for e in (exp_cov, exp_mean, exp_scale, exp_rot)]


def check_backward_chain_inria(got, g, scene, cam, w, h, ids, deg, clamped):
    """The upstream profile's per-Gaussian chain for the Gaussians `ids` against oracle/backward_np.py (inria_*), fed with
    the GPU's own upstream gradients. clamped: bool[N,3] of the forward call. Returns the largest expected magnitudes of
    (dL_dcov3D, dL_dmeans3D, dL_dscales, dL_drotations, dL_dshs)."""
    from oracle import backward_np as B
    fx, fy = w / (2.0 * cam.tan_fovx), h / (2.0 * cam.tan_fovy)
    m = len(ids)
    exp_cov, exp_mean, exp_scale, exp_rot, exp_sh = np.zeros((m, 6)), np.zeros((m, 3)), np.zeros((m, 3)), np.zeros((m, 4)), np.zeros((m, 48))
    for j, i in enumerate(ids):
        c3 = g["cov3D"][i].astype(np.float64)
        m3 = scene["means3D"][i, :3].astype(np.float64)
        dconic = got["dL_dconic_opacity"][i, :3].astype(np.float64)
        dcov = got["dL_dcov2D"][i].astype(np.float64) if "dL_dcov2D" in got else None
        exp_cov[j], g_mean_j = B.inria_conic_backward(c3, m3, cam.view, fx, fy, cam.tan_fovx, cam.tan_fovy, dconic, dcov)
        g_sh, g_mean_c = B.inria_color_backward(m3, cam.cam_pos, scene["shs"][i].reshape(16, 3), deg, got["dL_dcolors"][i].astype(np.float64))
        # the oracle decides the clamp from its own float64 colour; the kernel uses the forward's flags: they must agree
        raw_negative = B.inria_color(m3, cam.cam_pos, scene["shs"][i].reshape(16, 3), deg) == 0.0
        assert (raw_negative == clamped[i]).all() or np.abs(g["rgb"][i]).min() < 1e-6
        exp_sh[j] = g_sh.reshape(48)
        exp_mean[j] = (B.inria_project_mean2d_backward(m3, cam.proj, w, h, got["dL_dmean2D"][i].astype(np.float64)) + g_mean_j + g_mean_c)
        exp_scale[j], exp_rot[j] = B.inria_cov3d_backward(scene["scales"][i, :3], scene["rotations"][i], 1.0,
                                                          got["dL_dcov3D"][i].astype(np.float64))
    ids = np.asarray(ids)
    for name, e, gotv, tol in (("dL_dcov3D", exp_cov, got["dL_dcov3D"][ids], 2e-3),
                               ("dL_dmeans3D", exp_mean, got["dL_dmeans3D"][ids][:, :3], 3e-3),
                               ("dL_dscales", exp_scale, got["dL_dscales"][ids][:, :3], 3e-3),
                               ("dL_drotations", exp_rot, got["dL_drotations"][ids], 3e-3),
                               ("dL_dshs", exp_sh, got["dL_dshs"][ids], 1e-4)):
        if m == 0:
            continue
        err = np.abs(gotv - e).max(1)
        mag = np.maximum(np.abs(e).max(1), 1e-3 * np.abs(e).max())
        assert (err <= tol * np.maximum(mag, 1e-30)).all(), (name, float((err / np.maximum(mag, 1e-30)).max()))
    return [float(np.abs(e).max()) if m else 0.0 for e in (exp_cov, exp_mean, exp_scale, exp_rot, exp_sh)]
