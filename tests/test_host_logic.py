"""CPU: host-side pieces — camera prep, synthetic scenes, band partitioning."""
import math

import numpy as np

from gsrast_amd import camera, scenes, sharding


def test_default_camera_matches_the_glm_formulas():
    cam = camera.default_camera(1920, 1080, near=0.01, far=100.0)
    v = cam.view.reshape(4, 4).T        # [row, col]
    p = cam.proj.reshape(4, 4).T
    # eye (0,0,-5), looking +z, up (0,-1,0): s = (1,0,0), u = (0,-1,0), f = (0,0,1); row 2 negated
    assert np.allclose(v, [[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 5], [0, 0, 0, 1]], atol=1e-6)
    t = math.tan(math.radians(45) / 2)
    assert abs(p[0, 0] - 1 / (t * 1920 / 1080)) < 1e-5 and abs(p[1, 1] + 1 / t) < 1e-5
    assert abs(p[3, 2] - 1.0) < 1e-6 and abs(p[3, 3] - 5.0) < 1e-5          # clip.w = forward distance
    assert abs(cam.tan_fovy - t) < 1e-6 and abs(cam.tan_fovx - t * 1920 / 1080) < 1e-5


def test_scenes_are_reproducible_and_well_formed():
    a, b = scenes.isotropic_scene(500, 42), scenes.isotropic_scene(500, 42)
    for k in a:
        assert np.array_equal(a[k], b[k]) and a[k].dtype == np.float32
    assert a["means3D"].shape == (500, 4) and (a["means3D"][:, 3] == 1).all()
    assert np.allclose(a["scales"][:, 3], math.e) and (a["shs"][:, 3:] == 0).all()
    g = scenes.garden_like_scene(2000, 43)
    assert np.allclose(np.linalg.norm(g["rotations"], axis=1), 1.0, atol=1e-5)
    assert ((g["opacities"] > 0) & (g["opacities"] < 1)).all()
    # splitmix64 known answers (seed 0: first outputs of the reference generator)
    z = scenes.splitmix64(0, 0, 2)
    assert int(z[0]) == 0xE220A8397B1DCDAF and int(z[1]) == 0x6E789E6AA1B965F4


def test_trained_like_scene_has_flat_splats_on_surfaces_a_heavy_tail_and_a_background():
    """scenes.trained_like: what the thresholds of the library are checked on (scripts/thresholds_check.py). Its claims,
    checked against the oracle's covariance: the thin axis of a splat is 10-100 times thinner than the other two and the
    3-D covariance along it is its square; opacity is bimodal; 500 splats are huge and far out; reproducible."""
    from gsrast_amd import camera
    from oracle import cpu_oracle
    n = 40_000
    a, b = scenes.trained_like(n, 45), scenes.trained_like(n, 45)
    for k in a:
        assert np.array_equal(a[k], b[k]) and a[k].dtype == np.float32
    s3 = a["scales"][:, :3].astype(np.float64)
    ratio = np.minimum(s3[:, 0], s3[:, 1]) / s3[:, 2]
    assert (ratio > 9.9).all() and (ratio < 101.0).all()
    far_out = np.linalg.norm(a["means3D"][:, :3], axis=1) > 20.0
    assert far_out.sum() == 500 and (s3[far_out, :2] >= 2.0).all()
    tang = s3[~far_out, :2].max(axis=1)
    assert np.median(tang) < 0.03 and (tang > 0.2).mean() > 0.003           # log-normal body, heavy tail
    op = a["opacities"][~far_out]
    assert 0.45 < (op > 0.9).mean() < 0.7 and 0.2 < (op < 0.3).mean() < 0.4
    assert np.allclose(np.linalg.norm(a["rotations"], axis=1), 1.0, atol=1e-5)
    # the covariance the oracle builds (GSCuda.cu:168-195) along the quaternion's third axis is the thin scale squared
    cam = camera.default_camera(320, 192, near=0.05, far=200.0)
    cov = cpu_oracle.forward(a, cam, threads=4)["cov3D"].astype(np.float64)
    q = a["rotations"].astype(np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    v = np.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], axis=1)
    quad = (cov[:, 0] * v[:, 0] ** 2 + cov[:, 3] * v[:, 1] ** 2 + cov[:, 5] * v[:, 2] ** 2
            + 2 * (cov[:, 1] * v[:, 0] * v[:, 1] + cov[:, 2] * v[:, 0] * v[:, 2] + cov[:, 4] * v[:, 1] * v[:, 2]))
    seen = np.abs(cov).sum(axis=1) > 0                                       # (the reference writes cov3D inside the frustum only)
    assert seen.sum() > 1000
    # (float32 cancellation: the two in-surface terms are up to 10^4 times the thin one)
    assert np.median(np.abs(quad[seen] / s3[seen, 2] ** 2 - 1.0)) < 0.05


def test_band_partitions_cover_every_tile_row_once():
    for gy in (1, 7, 68, 135):
        for w in (1, 2, 4, 8):
            b = sharding.uniform_bands(gy, w)
            assert b[0] == 0 and b[-1] == gy and len(b) == w + 1 and all(x <= y for x, y in zip(b, b[1:]))
    cost = np.zeros(68)
    cost[20:40] = 100.0
    b = sharding.balanced_bands(cost, 4, floor_cost=1.0)
    assert b[0] == 0 and b[-1] == 68 and all(y > x for x, y in zip(b, b[1:]))
    loads = [cost[b[i]:b[i + 1]].sum() for i in range(4)]
    assert max(loads) <= 0.45 * cost.sum()
    assert sharding.balanced_bands(np.ones(3), 8) == [0, 1, 2, 3, 3, 3, 3, 3, 3]
