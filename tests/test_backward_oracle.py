"""CPU: the gradient oracle (oracle/backward_np.py) is pinned by central finite differences of the forward
functions it differentiates — the float64 restatement of the reference's blend loop (GSCuda.cu:623-676) and
of computeCov2D + conic (GSCuda.cu:197-231, :329-335). The reference has no backward pass to compare with."""
import numpy as np

from oracle import backward_np as B


def _toy(seed=0, n=14, w=24, h=20):
    rng = np.random.default_rng(seed)
    means = np.stack([rng.uniform(-2, w + 2, n), rng.uniform(-2, h + 2, n)], 1)
    # positive-definite conics of mixed size; a few nearly opaque splats so that alpha clamps and pixels terminate
    s = rng.uniform(0.02, 0.4, (n, 2))
    rho = rng.uniform(-0.6, 0.6, n)
    conic = np.stack([s[:, 0], rho * np.sqrt(s[:, 0] * s[:, 1]), s[:, 1]], 1)
    op = rng.uniform(0.05, 1.0, n)
    op[:3] = [1.0, 0.999, 0.97]
    conic[:3] *= 0.05
    co = np.concatenate([conic, op[:, None]], 1)
    col = rng.uniform(-0.2, 1.2, (n, 3))
    gx, gy = (w + 15) // 16, (h + 15) // 16
    order = rng.permutation(n)
    plist = np.tile(order, gx * gy).astype(np.int64)
    ranges = np.array([[t * n, (t + 1) * n] for t in range(gx * gy)], np.int64)
    g_out = rng.normal(size=(3, h, w))
    return means, co, col, ranges, plist, w, h, np.array([0.3, 0.1, 0.7]), g_out


def test_blend_gradients_match_finite_differences():
    means, co, col, ranges, plist, w, h, bg, g_out = _toy()
    out, ft, nc = B.blend_forward(means, co, col, ranges, plist, w, h, bg)
    assert nc.max() > 3 and (ft < 0.05).any() and (nc < len(means)).any()      # some pixels terminate early
    g = B.blend_backward(means, co, col, ranges, plist, nc, ft, w, h, bg, g_out)

    def loss_of(m=means, c=co, k=col):
        return float((B.blend_forward(m, c, k, ranges, plist, w, h, bg)[0] * g_out).sum())

    fd_mean = B.finite_difference(lambda x: loss_of(m=x), means, 1e-6)
    fd_co = B.finite_difference(lambda x: loss_of(c=x), co, 1e-7)
    fd_col = B.finite_difference(lambda x: loss_of(k=x), col, 1e-6)
    scale = max(1.0, np.abs(fd_mean).max())
    assert np.abs(g["dL_dmean2D"] - fd_mean).max() <= 2e-5 * scale
    assert np.abs(g["dL_dconic"] - fd_co[:, :3]).max() <= 2e-5 * max(1.0, np.abs(fd_co[:, :3]).max())
    assert np.abs(g["dL_dopacity"] - fd_co[:, 3]).max() <= 2e-5 * max(1.0, np.abs(fd_co[:, 3]).max())
    assert np.abs(g["dL_dcolor"] - fd_col).max() <= 1e-6 * max(1.0, np.abs(fd_col).max())
    # the clamped splat (opacity * G > 0.99 near its centre) still receives colour gradient
    assert np.abs(g["dL_dcolor"][0]).sum() > 0


def test_background_and_empty_tiles():
    means, co, col, ranges, plist, w, h, bg, g_out = _toy(seed=3)
    ranges = ranges.copy()
    ranges[1] = (0, 0)                                   # a tile nothing touches: out = background, no gradient
    out, ft, nc = B.blend_forward(means, co, col, ranges, plist, w, h, bg)
    assert np.allclose(out[:, 0:16, 16:24], bg[:, None, None]) and (nc[0:16, 16:24] == 0).all()
    g = B.blend_backward(means, co, col, ranges, plist, nc, ft, w, h, bg, g_out)
    fd = B.finite_difference(lambda x: float((B.blend_forward(means, co, x, ranges, plist, w, h, bg)[0] * g_out).sum()),
                             col, 1e-6)
    assert np.abs(g["dL_dcolor"] - fd).max() <= 1e-6 * max(1.0, np.abs(fd).max())


def test_cov3d_to_conic_gradient_matches_finite_differences():
    rng = np.random.default_rng(5)
    from gsrast_amd import camera
    cam = camera.default_camera(640, 480)
    focal = 480 / (2.0 * cam.tan_fovy)
    for _ in range(6):
        a = rng.normal(size=(3, 3)) * 0.2
        sig = a @ a.T + 0.01 * np.eye(3)
        c3 = np.array([sig[0, 0], sig[0, 1], sig[0, 2], sig[1, 1], sig[1, 2], sig[2, 2]])
        mean3 = rng.uniform(-1.5, 1.5, 3)
        gk = rng.normal(size=3)
        f = lambda x: float(B.cov2d_conic(x, mean3, cam.view, focal, cam.tan_fovx, cam.tan_fovy) @ gk)
        fd = B.finite_difference(f, c3, 1e-7)
        an = B.conic_backward(c3, mean3, cam.view, focal, cam.tan_fovx, cam.tan_fovy, gk)
        assert np.abs(an - fd).max() <= 1e-5 * max(1.0, np.abs(fd).max()), (an, fd)


def test_input_chain_gradients_match_finite_differences():
    """means3D -> pixel centre, (scale, rotation) -> cov3D, means3D -> conic (through the Jacobian)."""
    rng = np.random.default_rng(11)
    from gsrast_amd import camera
    cam = camera.default_camera(640, 480)
    focal = 480 / (2.0 * cam.tan_fovy)
    for trial in range(8):
        mean3 = rng.uniform(-1.5, 1.5, 3)
        if trial >= 6:
            mean3 = np.array([9.0, -7.0, 0.5])            # far off axis: t.x / t.z and t.y / t.z are clamped
        scale, rot = np.exp(rng.uniform(-3, -1, 3)), rng.normal(size=4)
        g2, g6, gk = rng.normal(size=2), rng.normal(size=6), rng.normal(size=3)
        fd = B.finite_difference(lambda m: float(B.project_mean2d(m, cam.proj, 640, 480) @ g2), mean3, 1e-6)
        an = B.project_mean2d_backward(mean3, cam.proj, 640, 480, g2)
        assert np.abs(an - fd).max() <= 1e-6 * max(1.0, np.abs(fd).max())
        f = lambda s, r: float(B.cov3d(s, r, 1.3) @ g6)
        gs, gr = B.cov3d_backward(scale, rot, 1.3, g6)
        assert np.abs(gs - B.finite_difference(lambda s: f(s, rot), scale, 1e-7)).max() <= 1e-6 * max(1.0, np.abs(gs).max())
        assert np.abs(gr - B.finite_difference(lambda r: f(scale, r), rot, 1e-7)).max() <= 1e-6 * max(1.0, np.abs(gr).max())
        c3 = B.cov3d(scale, rot)
        fdm = B.finite_difference(lambda m: float(B.cov2d_conic(c3, m, cam.view, focal, cam.tan_fovx, cam.tan_fovy) @ gk),
                                  mean3, 1e-6)
        anm = B.conic_backward_mean(c3, mean3, cam.view, focal, cam.tan_fovx, cam.tan_fovy, gk)
        assert np.abs(anm - fdm).max() <= 2e-5 * max(1.0, np.abs(fdm).max()), (trial, anm, fdm)


def test_vectorised_tile_backward_matches_the_per_pixel_loops():
    """blend_tile_backward (what the full-size GPU test of BASELINE config 5 uses) against blend_forward / blend_backward."""
    for seed in (0, 3, 5):
        means, co, col, ranges, plist, w, h, bg, g_out = _toy(seed=seed, n=40, w=40, h=30)
        co = co.copy()
        co[:6, :3] *= 0.3                                   # enough coverage that some pixels terminate
        out, ft, nc = B.blend_forward(means, co, col, ranges, plist, w, h, bg)
        g = B.blend_backward(means, co, col, ranges, plist, nc, ft, w, h, bg, g_out)
        gx, gy = (w + 15) // 16, (h + 15) // 16
        acc = {k: np.zeros_like(v) for k, v in g.items()}
        for ty in range(gy):
            for tx in range(gx):
                a, b = ranges[ty * gx + tx]
                ids = plist[a:b]
                tile_g = np.zeros((3, 16, 16))
                y1, x1 = min(h, ty * 16 + 16), min(w, tx * 16 + 16)
                tile_g[:, : y1 - ty * 16, : x1 - tx * 16] = g_out[:, ty * 16:y1, tx * 16:x1]
                r = B.blend_tile_backward(means[ids], co[ids], col[ids], tx, ty, w, h, bg, tile_g)
                np.add.at(acc["dL_dmean2D"], ids, r["d_mean"])
                np.add.at(acc["dL_dconic"], ids, r["d_conic"])
                np.add.at(acc["dL_dopacity"], ids, r["d_op"])
                np.add.at(acc["dL_dcolor"], ids, r["d_col"])
                # the gradient w.r.t. the 2-D covariance, summed pixel by pixel as 0.5 dLp u u^T (what the HIP render backward
                # does), is -K (dL/dK) K of the summed conic gradient, K the conic as a full symmetric matrix
                for j in range(len(ids)):
                    K = np.array([[co[ids[j], 0], co[ids[j], 1]], [co[ids[j], 1], co[ids[j], 2]]])
                    gK = np.array([[r["d_conic"][j, 0], 0.5 * r["d_conic"][j, 1]], [0.5 * r["d_conic"][j, 1], r["d_conic"][j, 2]]])
                    m = -K @ gK @ K
                    assert np.abs(r["d_cov"][j] - np.array([m[0, 0], m[0, 1], m[1, 1]])).max() <= 1e-9 * max(1e-6, np.abs(m).max()), (tx, ty, j)
                assert np.abs(r["out"][:, : y1 - ty * 16, : x1 - tx * 16] - out[:, ty * 16:y1, tx * 16:x1]).max() <= 1e-12
                assert np.array_equal(r["n_contrib"][: y1 - ty * 16, : x1 - tx * 16], nc[ty * 16:y1, tx * 16:x1])
                assert np.abs(r["final_t"][: y1 - ty * 16, : x1 - tx * 16] - ft[ty * 16:y1, tx * 16:x1]).max() <= 1e-12
        assert (nc < 40).any() and (ft < 0.01).any()
        for k in g:
            assert np.abs(acc[k] - g[k]).max() <= 1e-9 * max(1.0, np.abs(g[k]).max()), k


def test_upstream_profile_chain_matches_finite_differences():
    """The per-Gaussian functions of the `inria` profile (SH colour with its view direction and clamp, raw-quaternion
    covariance, two focal lengths, its pixel-centre formula) against central differences."""
    rng = np.random.default_rng(7)
    view = np.array([0.96, 0.1, -0.26, 0, -0.05, 0.98, 0.19, 0, 0.27, -0.17, 0.95, 0, 0.3, -0.2, 4.0, 1.0])
    proj = np.array([1.7, 0.1, 0.3, 0.28, -0.1, 2.4, 0.2, -0.17, 0.3, -0.2, 1.0, 0.95, 0.4, -0.3, 3.8, 4.0])
    fx, fy, tx_, ty_ = 900.0, 1150.0, 0.9, 0.5
    for trial in range(6):
        mean = rng.uniform(-1.5, 1.5, 3)
        if trial == 4:
            mean = np.array([6.0, 0.3, 1.0])                 # far off axis: the +-1.3 tan(fov) clamp engages
        cam = rng.uniform(-3, 3, 3)
        sh = rng.normal(0, 0.5, (16, 3))
        if trial == 2:
            sh[0] = (-6.0, 0.2, 0.1)                         # red channel clamped at zero
        g_col = rng.normal(size=3)
        for deg in (0, 1, 2, 3):
            col = B.inria_color(mean, cam, sh, deg)
            g_sh, g_mean = B.inria_color_backward(mean, cam, sh, deg, g_col)
            fd_sh = B.finite_difference(lambda s_: float(B.inria_color(mean, cam, s_, deg) @ g_col), sh, 1e-6)
            fd_mean = B.finite_difference(lambda m_: float(B.inria_color(m_, cam, sh, deg) @ g_col), mean, 1e-6)
            assert np.abs(g_sh - fd_sh).max() <= 1e-6 * max(1.0, np.abs(fd_sh).max())
            assert np.abs(g_mean - fd_mean).max() <= 2e-6 * max(1.0, np.abs(fd_mean).max())
            if trial == 2:
                assert col[0] == 0.0 and (g_sh[:, 0] == 0).all()
            assert (g_sh[(deg + 1) ** 2:] == 0).all()
        scale = np.exp(rng.uniform(-3, -0.5, 3))
        quat = rng.normal(size=4)                            # not a unit quaternion: the forward does not normalise
        c3 = B.inria_cov3d(scale, quat, 1.3)
        g6 = rng.normal(size=6)
        g_scale, g_rot = B.inria_cov3d_backward(scale, quat, 1.3, g6)
        fd_s = B.finite_difference(lambda s_: float(B.inria_cov3d(s_, quat, 1.3) @ g6), scale, 1e-7)
        fd_q = B.finite_difference(lambda q_: float(B.inria_cov3d(scale, q_, 1.3) @ g6), quat, 1e-6)
        assert np.abs(g_scale - fd_s).max() <= 1e-5 * max(1e-3, np.abs(fd_s).max())
        assert np.abs(g_rot - fd_q).max() <= 1e-5 * max(1e-3, np.abs(fd_q).max())
        gk = rng.normal(size=3)
        g_c3, g_mean_j = B.inria_conic_backward(c3, mean, view, fx, fy, tx_, ty_, gk)
        fd_c3 = B.finite_difference(lambda c_: float(B.inria_cov2d_conic(c_, mean, view, fx, fy, tx_, ty_) @ gk), c3, 1e-7)
        fd_mj = B.finite_difference(lambda m_: float(B.inria_cov2d_conic(c3, m_, view, fx, fy, tx_, ty_) @ gk), mean, 1e-6)
        assert np.abs(g_c3 - fd_c3).max() <= 2e-5 * max(1e-6, np.abs(fd_c3).max())
        assert np.abs(g_mean_j - fd_mj).max() <= 2e-5 * max(1e-6, np.abs(fd_mj).max())
        g2 = rng.normal(size=2)
        g_m = B.inria_project_mean2d_backward(mean, proj, 1920, 1080, g2)
        fd_m = B.finite_difference(lambda m_: float(B.inria_project_mean2d(m_, proj, 1920, 1080) @ g2), mean, 1e-6)
        assert np.abs(g_m - fd_m).max() <= 1e-6 * max(1.0, np.abs(fd_m).max())
    # and the two profiles agree where they coincide: a unit quaternion gives the same covariance
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    assert np.abs(B.inria_cov3d(scale, q, 1.0) - B.cov3d(scale, q, 1.0)).max() <= 1e-12
