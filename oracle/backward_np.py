"""Gradient oracle of the forward rasterizer (BASELINE config 5), numpy float64.

TEST INFRASTRUCTURE: imported by tests/ only, never by the product path.

**Parity unpinned**: the reference has no backward pass (apps/gsrast/gscuda has none, and the
upstream submodule that has one is an empty directory), so there is no reference gradient to
compare with. What pins this oracle instead is the forward function itself: `blend_forward` restates
the reference's blend loop (apps/gsrast/gscuda/GSCuda.cu:623-676) and `cov2d_conic` its
computeCov2D / conic (GSCuda.cu:197-231, :329-335) in float64; the analytic gradients below are
checked against central finite differences of exactly those functions (tests/test_backward_oracle.py).

Conventions (also those of gsr_backward in include/gsrast_amd.h):
  L = sum over pixels and channels of dL_dout * out_color
  dL_dmean2D  [N,2]  w.r.t. the pixel-space centre means2D (the reference's pixel units, no NDC factor)
  dL_dconic   [N,3]  w.r.t. the three stored numbers (A, B, C) of power = -0.5 (A dx^2 + C dy^2) - B dx dy
  dL_dopacity [N]    w.r.t. conicOpacity.w
  dL_dcolor   [N,3]
  dL_dcov3D   [N,6]  w.r.t. the six stored numbers of the symmetric 3-D covariance (GSCuda.cu:189-194)
  dL_ddc      [N,3]  w.r.t. the DC spherical-harmonic coefficients (colour = 0.5 + 0.4 DC, GSCuda.cu:362-366)
The forward function is piecewise smooth: the hard tests (power > 0, alpha < 1/255, alpha clamp at 0.99,
transmittance cut-off) select a branch and contribute no derivative of their own; where alpha is clamped
to 0.99 its derivative w.r.t. the Gaussian's parameters is zero.
"""
from __future__ import annotations

import numpy as np

TILE = 16


def _pixel_lists(ranges, point_list, width, height):
    gx = (width + TILE - 1) // TILE
    for py in range(height):
        for px in range(width):
            t = (py // TILE) * gx + (px // TILE)
            a, b = int(ranges[t, 0]), int(ranges[t, 1])
            yield px, py, point_list[a:b] if b > a else point_list[0:0]


def blend_forward(means2D, conic_opacity, colors, ranges, point_list, width, height, background, t_cutoff=0.001):
    """GSCuda.cu:623-676 per pixel, float64. Returns out[3,H,W], finalT[H,W], n_contrib[H,W]."""
    m = np.asarray(means2D, np.float64)
    co = np.asarray(conic_opacity, np.float64)
    col = np.asarray(colors, np.float64)
    bg = np.asarray(background, np.float64)
    out = np.zeros((3, height, width))
    final_t = np.ones((height, width))
    n_contrib = np.zeros((height, width), np.int64)
    for px, py, ids in _pixel_lists(ranges, point_list, width, height):
        T, C, last = 1.0, np.zeros(3), 0
        for k, g in enumerate(ids):
            dx, dy = m[g, 0] - px, m[g, 1] - py
            power = -0.5 * (co[g, 0] * dx * dx + co[g, 2] * dy * dy) - co[g, 1] * dx * dy
            if power > 0.0:
                continue
            alpha = min(0.99, co[g, 3] * np.exp(power))
            if alpha < 1.0 / 255.0:
                continue
            test = T * (1.0 - alpha)
            if test < t_cutoff:
                break
            C = C + col[g] * alpha * T
            T = test
            last = k + 1
        out[:, py, px] = C + T * bg
        final_t[py, px] = T
        n_contrib[py, px] = last
    return out, final_t, n_contrib


def blend_backward(means2D, conic_opacity, colors, ranges, point_list, n_contrib, final_t, width, height, background,
                   dL_dout):
    """Analytic gradients of sum(dL_dout * out) through the blend loop, walking each pixel's list back to
    front from its last contributor (the quantities a GPU backward pass has: nContrib, finalT)."""
    m = np.asarray(means2D, np.float64)
    co = np.asarray(conic_opacity, np.float64)
    col = np.asarray(colors, np.float64)
    bg = np.asarray(background, np.float64)
    g_out = np.asarray(dL_dout, np.float64)
    n = m.shape[0]
    d_mean, d_conic, d_op, d_col = np.zeros((n, 2)), np.zeros((n, 3)), np.zeros(n), np.zeros((n, 3))
    for px, py, ids in _pixel_lists(ranges, point_list, width, height):
        gp = g_out[:, py, px]
        T = float(final_t[py, px])
        S = T * float(bg @ gp)                 # what lies behind the current record, dotted with dL/dC
        for k in range(int(n_contrib[py, px]) - 1, -1, -1):
            g = ids[k]
            dx, dy = m[g, 0] - px, m[g, 1] - py
            A, B, Cc, op = co[g]
            power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
            if power > 0.0:
                continue
            G = np.exp(power)
            raw = op * G
            alpha = min(0.99, raw)
            if alpha < 1.0 / 255.0:
                continue
            T = T / (1.0 - alpha)              # transmittance in front of this record
            cg = float(col[g] @ gp)
            d_col[g] += alpha * T * gp
            dL_dalpha = T * cg - S / (1.0 - alpha)
            S += cg * alpha * T
            if raw > 0.99:                     # clamped: alpha does not move with the Gaussian's parameters
                continue
            d_op[g] += G * dL_dalpha
            dL_dpower = op * G * dL_dalpha
            d_conic[g, 0] += -0.5 * dx * dx * dL_dpower
            d_conic[g, 1] += -dx * dy * dL_dpower
            d_conic[g, 2] += -0.5 * dy * dy * dL_dpower
            d_mean[g, 0] += (-A * dx - B * dy) * dL_dpower
            d_mean[g, 1] += (-Cc * dy - B * dx) * dL_dpower
    return {"dL_dmean2D": d_mean, "dL_dconic": d_conic, "dL_dopacity": d_op, "dL_dcolor": d_col}


def blend_tile_backward(xy, co, col, tx, ty, width, height, background, dL_dout_tile, t_cutoff=0.001, f32_forward=False):
    """One 16 x 16 tile, vectorised [L records x 256 pixels], float64: the forward loop of `blend_forward` and the
    gradients of `blend_backward` restricted to this tile's pixels. Used at frame sizes where the per-pixel
    Python loops above would take hours; tests/test_backward_oracle.py checks it against them.

    xy [L,2], co [L,4], col [L,3]: the records of the tile's list, front to back (a prefix that reaches the last
    contributor of every pixel is enough). dL_dout_tile [3,16,16] (entries outside the image are ignored).
    Returns dict: per-record sums d_mean [L,2], d_conic [L,3], d_cov [L,3] (w.r.t. the full symmetric 2-D covariance: m00, m01,
    m11 — every pixel's share of dL/dK is -0.5 dLp d d^T, so its share of -K (dL/dK) K is 0.5 dLp u u^T with u = K d),
    d_op [L], d_col [L,3]; per pixel out [3,16,16],
    final_t [16,16], n_contrib [16,16] (pixels outside the image: 0).
    f32_forward: power, exp (libm's expf), alpha and the transmittance product are evaluated in float32 in the forward's
    operation order (GSCuda.cu:634-657) — the function the HIP backward differentiates, decisions included: with terms of
    1e3 in the power of a screen-filling splat a float64 forward is another function at the 1e-4 level — and the gradient
    sums over them in float64: what remains against the HIP backward is the backward's own arithmetic."""
    xy = np.asarray(xy, np.float64).reshape(-1, 2)
    co = np.asarray(co, np.float64).reshape(-1, 4)
    col = np.asarray(col, np.float64).reshape(-1, 3)
    bg = np.asarray(background, np.float64)
    L = xy.shape[0]
    ys, xs = np.mgrid[0:TILE, 0:TILE]
    px = (tx * TILE + xs).reshape(-1).astype(np.float64)
    py = (ty * TILE + ys).reshape(-1).astype(np.float64)
    inside = (px < width) & (py < height)
    gp = np.asarray(dL_dout_tile, np.float64).reshape(3, -1) * inside[None, :]
    if L == 0:
        out = np.where(inside[None, :], bg[:, None], 0.0)
        return {"d_mean": np.zeros((0, 2)), "d_conic": np.zeros((0, 3)), "d_cov": np.zeros((0, 3)), "d_op": np.zeros(0), "d_col": np.zeros((0, 3)),
                "out": out.reshape(3, TILE, TILE), "final_t": inside.astype(np.float64).reshape(TILE, TILE),
                "n_contrib": np.zeros((TILE, TILE), np.int64)}
    dx = xy[:, 0:1] - px[None, :]
    dy = xy[:, 1:2] - py[None, :]
    A, B, Cc, op = co[:, 0:1], co[:, 1:2], co[:, 2:3], co[:, 3:4]
    if f32_forward:
        from oracle import cpu_oracle
        f = np.float32
        dx32, dy32 = f(xy[:, 0:1]) - f(px[None, :]), f(xy[:, 1:2]) - f(py[None, :])
        A32, B32, C32, op32 = f(A), f(B), f(Cc), f(op)
        power = f(-0.5) * ((A32 * dx32) * dx32 + (C32 * dy32) * dy32) - (B32 * dx32) * dy32
        G = cpu_oracle.expf(np.minimum(power, f(0.0))).reshape(power.shape)
        raw = op32 * G
        alpha = np.minimum(f(0.99), raw)
        valid = (power <= 0.0) & (alpha >= f(1.0 / 255.0)) & inside[None, :]
        a_eff = np.where(valid, alpha, f(0.0))
        t_after = np.cumprod(f(1.0) - a_eff, axis=0, dtype=np.float32).astype(np.float64)
        power, G, raw, alpha, a_eff = (v.astype(np.float64) for v in (power, G, raw, alpha, a_eff))
    else:
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        with np.errstate(over="ignore"):
            G = np.exp(np.minimum(power, 0.0))
        raw = op * G
        alpha = np.minimum(0.99, raw)
        valid = (power <= 0.0) & (alpha >= 1.0 / 255.0) & inside[None, :]
        a_eff = np.where(valid, alpha, 0.0)
        t_after = np.cumprod(1.0 - a_eff, axis=0)                   # transmittance behind record k (no termination yet)
    t_before = np.vstack([np.ones((1, px.size)), t_after[:-1]])
    stop = valid & (t_after < t_cutoff)                             # GSCuda.cu:653: this record ends the pixel
    has_stop = stop.any(axis=0)
    stop_idx = np.where(has_stop, stop.argmax(axis=0), L)
    k = np.arange(L)[:, None]
    contrib = valid & (k < stop_idx[None, :])
    final_t = np.where(has_stop, t_before[np.minimum(stop_idx, L - 1), np.arange(px.size)], t_after[-1])
    w = np.where(contrib, alpha * t_before, 0.0)                    # weight of record k in the pixel
    out = (col.T[:, :, None] * w[None]).sum(axis=1) + final_t[None, :] * bg[:, None]
    n_contrib = np.where(contrib.any(axis=0), L - np.argmax(contrib[::-1], axis=0), 0)
    # back to front: S_k = (everything behind record k) . dL/dC of the pixel
    cg = col @ gp                                                   # [L, 256]
    wc = w * cg
    behind = (final_t * (bg @ gp))[None, :] + (wc.sum(axis=0)[None, :] - np.cumsum(wc, axis=0))
    dL_dalpha = np.where(contrib, t_before * cg - behind / (1.0 - np.where(contrib, alpha, 0.0)), 0.0)
    d_col = np.einsum("lp,cp->lc", w, gp)
    free = contrib & ~(raw > 0.99)                                  # clamped alpha does not move with the parameters
    dLa = np.where(free, dL_dalpha, 0.0)
    d_op = (G * dLa).sum(axis=1)
    dLp = op * G * dLa
    d_conic = np.stack([(-0.5 * dx * dx * dLp).sum(1), (-dx * dy * dLp).sum(1), (-0.5 * dy * dy * dLp).sum(1)], 1)
    d_mean = np.stack([((-A * dx - B * dy) * dLp).sum(1), ((-Cc * dy - B * dx) * dLp).sum(1)], 1)
    ux, uy = A * dx + B * dy, B * dx + Cc * dy
    d_cov = np.stack([(0.5 * ux * ux * dLp).sum(1), (0.5 * ux * uy * dLp).sum(1), (0.5 * uy * uy * dLp).sum(1)], 1)
    return {"d_mean": d_mean, "d_conic": d_conic, "d_cov": d_cov, "d_op": d_op, "d_col": d_col,
            "out": np.where(inside[None, :], out, 0.0).reshape(3, TILE, TILE),
            "final_t": np.where(inside, final_t, 0.0).reshape(TILE, TILE), "n_contrib": n_contrib.reshape(TILE, TILE)}


# ---- cov3D -> conic (computeCov2D + inverse), and its gradient ------------------------------------
def _jw(mean3, view, focal, tan_fovx, tan_fovy):
    """The 2 x 3 matrix P with cov2D = P Sigma P^T (GSCuda.cu:201-225). view: 16 floats, column-major."""
    v = np.asarray(view, np.float64)
    t = np.array([v[0] * mean3[0] + v[4] * mean3[1] + v[8] * mean3[2] + v[12],
                  v[1] * mean3[0] + v[5] * mean3[1] + v[9] * mean3[2] + v[13],
                  v[2] * mean3[0] + v[6] * mean3[1] + v[10] * mean3[2] + v[14]])
    limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
    tx = min(limx, max(-limx, t[0] / t[2])) * t[2]
    ty = min(limy, max(-limy, t[1] / t[2])) * t[2]
    J = np.array([[focal / t[2], 0.0, -focal * tx / (t[2] * t[2])],
                  [0.0, focal / t[2], -focal * ty / (t[2] * t[2])]])
    W = np.array([[v[0], v[4], v[8]], [v[1], v[5], v[9]], [v[2], v[6], v[10]]])   # rotation rows of the view matrix
    return J @ W


def _sigma(c3):
    return np.array([[c3[0], c3[1], c3[2]], [c3[1], c3[3], c3[4]], [c3[2], c3[4], c3[5]]], np.float64)


def cov2d_conic(c3, mean3, view, focal, tan_fovx, tan_fovy):
    """(A, B, C) of the conic from the 6 stored covariance numbers (GSCuda.cu:197-231, :329-335)."""
    P = _jw(mean3, view, focal, tan_fovx, tan_fovy)
    cov = P @ _sigma(c3) @ P.T
    a, b, c = cov[0, 0] + 0.3, cov[0, 1], cov[1, 1] + 0.3
    det = a * c - b * b
    return np.array([c / det, -b / det, a / det])


def _gcov(K, dL_dconic, dL_dcov2D):
    """Gradient w.r.t. the full symmetric cov2D: given directly as (m00, m01, m11), or -K gK K from dL/d(A, B, C)."""
    if dL_dcov2D is not None:
        m00, m01, m11 = (float(v) for v in dL_dcov2D[:3])
        return np.array([[m00, m01], [m01, m11]])
    gA, gB, gC = dL_dconic
    return -K @ np.array([[gA, 0.5 * gB], [0.5 * gB, gC]]) @ K        # d(M^-1) = -M^-1 dM M^-1


def conic_backward(c3, mean3, view, focal, tan_fovx, tan_fovy, dL_dconic, dL_dcov2D=None):
    """dL/d(6 covariance numbers) from dL/d(A, B, C) — or from dL/dcov2D (m00, m01, m11) when that is given."""
    P = _jw(mean3, view, focal, tan_fovx, tan_fovy)
    cov = P @ _sigma(c3) @ P.T
    a, b, c = cov[0, 0] + 0.3, cov[0, 1], cov[1, 1] + 0.3
    det = a * c - b * b
    K = np.array([[c, -b], [-b, a]]) / det
    gM = _gcov(K, dL_dconic, dL_dcov2D)
    gS = P.T @ gM @ P                                       # w.r.t. the full 3 x 3 Sigma entries
    return np.array([gS[0, 0], 2.0 * gS[0, 1], 2.0 * gS[0, 2], gS[1, 1], 2.0 * gS[1, 2], gS[2, 2]])


def finite_difference(f, x, eps):
    """Central differences of the scalar function f at x (flattened), one coordinate at a time."""
    x = np.array(x, np.float64)
    g = np.zeros_like(x)
    flat, gf = x.reshape(-1), g.reshape(-1)
    for i in range(flat.size):
        old = flat[i]
        flat[i] = old + eps
        hi = f(x)
        flat[i] = old - eps
        lo = f(x)
        flat[i] = old
        gf[i] = (hi - lo) / (2.0 * eps)
    return g


# ---- the per-Gaussian chain down to the inputs: means3D, scales, rotations ---------------------------
# Forward functions restated in float64 from the reference (and from csrc/preprocess.hip, which keeps
# its operation order): projection to the pixel-space centre GSCuda.cu:302-305, :342; quatToMat
# :157-162; computeCov3D :168-195; computeCov2D :197-231.
def project_mean2d(mean3, proj, width, height):
    p = np.asarray(proj, np.float64)
    hx = p[0] * mean3[0] + p[4] * mean3[1] + p[8] * mean3[2] + p[12]
    hy = p[1] * mean3[0] + p[5] * mean3[1] + p[9] * mean3[2] + p[13]
    hw = p[3] * mean3[0] + p[7] * mean3[1] + p[11] * mean3[2] + p[15]
    ow = 1.0 / (0.001 + hw)
    return np.array([(hx * ow * 0.5 + 0.5) * width, (hy * ow * 0.5 + 0.5) * height])


def project_mean2d_backward(mean3, proj, width, height, g2):
    p = np.asarray(proj, np.float64)
    hx = p[0] * mean3[0] + p[4] * mean3[1] + p[8] * mean3[2] + p[12]
    hy = p[1] * mean3[0] + p[5] * mean3[1] + p[9] * mean3[2] + p[13]
    wp = 0.001 + (p[3] * mean3[0] + p[7] * mean3[1] + p[11] * mean3[2] + p[15])
    g = np.zeros(3)
    for j in range(3):
        dx = 0.5 * width * (p[4 * j + 0] / wp - hx * p[4 * j + 3] / (wp * wp))
        dy = 0.5 * height * (p[4 * j + 1] / wp - hy * p[4 * j + 3] / (wp * wp))
        g[j] = dx * g2[0] + dy * g2[1]
    return g


def quat_to_mat(q):
    """Rotation matrix (math rows / columns) of the normalised quaternion, real part first (GSCuda.cu:157-162)."""
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[2 * (x * x + y * y) - 1, 2 * (y * z - x * w), 2 * (y * w + x * z)],
                     [2 * (y * z + x * w), 2 * (x * x + z * z) - 1, 2 * (z * w - x * y)],
                     [2 * (y * w - x * z), 2 * (z * w + x * y), 2 * (x * x + w * w) - 1]])


def cov3d(scale, rot, mod=1.0):
    m = quat_to_mat(rot) @ np.diag(mod * np.asarray(scale, np.float64)[:3])
    s = m @ m.T
    return np.array([s[0, 0], s[0, 1], s[0, 2], s[1, 1], s[1, 2], s[2, 2]])


def cov3d_backward(scale, rot, mod, g6):
    """dL/dscale[3], dL/drot[4] (w.r.t. the un-normalised quaternion) from dL/d(6 covariance numbers)."""
    q = np.asarray(rot, np.float64)
    n = np.linalg.norm(q)
    x, y, z, w = q / n
    R = quat_to_mat(rot)
    s = mod * np.asarray(scale, np.float64)[:3]
    M = R @ np.diag(s)
    gS = np.array([[g6[0], 0.5 * g6[1], 0.5 * g6[2]], [0.5 * g6[1], g6[3], 0.5 * g6[4]], [0.5 * g6[2], 0.5 * g6[4], g6[5]]])
    gM = 2.0 * gS @ M
    g_scale = mod * (R * gM).sum(0)
    gR = gM * s[None, :]
    dR = {"x": np.array([[4 * x, -2 * w, 2 * z], [2 * w, 4 * x, -2 * y], [-2 * z, 2 * y, 4 * x]]),
          "y": np.array([[4 * y, 2 * z, 2 * w], [2 * z, 0, -2 * x], [2 * w, 2 * x, 0]]),
          "z": np.array([[0, 2 * y, 2 * x], [2 * y, 4 * z, 2 * w], [-2 * x, 2 * w, 0]]),
          "w": np.array([[0, -2 * x, 2 * y], [2 * x, 0, 2 * z], [2 * y, 2 * z, 4 * w]])}
    gq_hat = np.array([(gR * dR[k]).sum() for k in "xyzw"])
    qh = q / n
    return g_scale, (gq_hat - qh * (qh @ gq_hat)) / n


def conic_backward_mean(c3, mean3, view, focal, tan_fovx, tan_fovy, dL_dconic, dL_dcov2D=None):
    """dL/dmean3 through the dependence of cov2D on the view-space position t (the Jacobian J)."""
    v = np.asarray(view, np.float64)
    t = np.array([v[0] * mean3[0] + v[4] * mean3[1] + v[8] * mean3[2] + v[12],
                  v[1] * mean3[0] + v[5] * mean3[1] + v[9] * mean3[2] + v[13],
                  v[2] * mean3[0] + v[6] * mean3[1] + v[10] * mean3[2] + v[14]])
    limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
    rx, ry = t[0] / t[2], t[1] / t[2]
    cx, cy = min(limx, max(-limx, rx)), min(limy, max(-limy, ry))
    tx, ty, tz = cx * t[2], cy * t[2], t[2]
    J = np.array([[focal / tz, 0.0, -focal * tx / (tz * tz)], [0.0, focal / tz, -focal * ty / (tz * tz)]])
    W = np.array([[v[0], v[4], v[8]], [v[1], v[5], v[9]], [v[2], v[6], v[10]]])
    Mw = W @ _sigma(c3) @ W.T
    cov = J @ Mw @ J.T
    a, b, c = cov[0, 0] + 0.3, cov[0, 1], cov[1, 1] + 0.3
    det = a * c - b * b
    K = np.array([[c, -b], [-b, a]]) / det
    gcov = _gcov(K, dL_dconic, dL_dcov2D)
    gJ = 2.0 * gcov @ J @ Mw
    g_tx = gJ[0, 2] * (-focal / (tz * tz))
    g_ty = gJ[1, 2] * (-focal / (tz * tz))
    g_tz = (gJ[0, 0] + gJ[1, 1]) * (-focal / (tz * tz)) + (gJ[0, 2] * tx + gJ[1, 2] * ty) * (2.0 * focal / (tz ** 3))
    clamped_x, clamped_y = rx != cx, ry != cy
    gt = np.array([0.0 if clamped_x else g_tx, 0.0 if clamped_y else g_ty,
                   g_tz + (g_tx * cx if clamped_x else 0.0) + (g_ty * cy if clamped_y else 0.0)])
    return W.T @ gt


# ---- the upstream (`inria`) semantics profile: per-Gaussian forward functions and their gradients ----------------
# Forward restated in float64 from csrc/preprocess_inria.hip / oracle/inria_np.py (SURVEY.md divergence table D1-D9):
# real SH up to degree 3 with + 0.5 and a clamp at zero, raw (not re-normalised) quaternion, separate focal lengths,
# w epsilon 1e-7, pixel centre ((ndc + 1) S - 1) / 2. Parity unpinned (no upstream source in the tree); the analytic
# gradients are checked against central finite differences of these same functions (tests/test_backward_oracle.py).
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def sh_basis(d, deg):
    """The 16 basis values B_k at the unit direction d (zeros beyond `deg`) and their gradients dB_k/dd [16,3]."""
    x, y, z = d
    B = np.zeros(16)
    G = np.zeros((16, 3))
    B[0] = SH_C0
    if deg > 0:
        B[1], B[2], B[3] = -SH_C1 * y, SH_C1 * z, -SH_C1 * x
        G[1, 1], G[2, 2], G[3, 0] = -SH_C1, SH_C1, -SH_C1
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        B[4:9] = (SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy))
        G[4] = (SH_C2[0] * y, SH_C2[0] * x, 0.0)
        G[5] = (0.0, SH_C2[1] * z, SH_C2[1] * y)
        G[6] = (-2 * SH_C2[2] * x, -2 * SH_C2[2] * y, 4 * SH_C2[2] * z)
        G[7] = (SH_C2[3] * z, 0.0, SH_C2[3] * x)
        G[8] = (2 * SH_C2[4] * x, -2 * SH_C2[4] * y, 0.0)
        if deg > 2:
            B[9:16] = (SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
                       SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                       SH_C3[6] * x * (xx - 3 * yy))
            G[9] = (SH_C3[0] * 6 * xy, SH_C3[0] * (3 * xx - 3 * yy), 0.0)
            G[10] = (SH_C3[1] * yz, SH_C3[1] * xz, SH_C3[1] * xy)
            G[11] = (SH_C3[2] * -2 * xy, SH_C3[2] * (4 * zz - xx - 3 * yy), SH_C3[2] * 8 * yz)
            G[12] = (SH_C3[3] * -6 * xz, SH_C3[3] * -6 * yz, SH_C3[3] * (6 * zz - 3 * xx - 3 * yy))
            G[13] = (SH_C3[4] * (4 * zz - 3 * xx - yy), SH_C3[4] * -2 * xy, SH_C3[4] * 8 * xz)
            G[14] = (SH_C3[5] * 2 * xz, SH_C3[5] * -2 * yz, SH_C3[5] * (xx - yy))
            G[15] = (SH_C3[6] * (3 * xx - 3 * yy), SH_C3[6] * -6 * xy, 0.0)
    return B, G


def inria_color(mean3, cam_pos, sh, deg):
    """colour[3] = max(0, 0.5 + sum_k B_k(dir) sh[k]) with dir = (mean - cam) / |mean - cam|; sh: [16,3]."""
    v = np.asarray(mean3, np.float64) - np.asarray(cam_pos, np.float64)
    B, _ = sh_basis(v / np.linalg.norm(v), deg)
    return np.maximum(B @ np.asarray(sh, np.float64).reshape(16, 3) + 0.5, 0.0)


def inria_color_backward(mean3, cam_pos, sh, deg, g_col):
    """(dL/dsh [16,3], dL/dmean3 [3]) from dL/dcolour[3]; a channel clamped at zero passes no gradient."""
    sh = np.asarray(sh, np.float64).reshape(16, 3)
    v = np.asarray(mean3, np.float64) - np.asarray(cam_pos, np.float64)
    ln = np.linalg.norm(v)
    d = v / ln
    B, G = sh_basis(d, deg)
    raw = B @ sh + 0.5
    g = np.where(raw < 0.0, 0.0, np.asarray(g_col, np.float64))
    g_sh = np.outer(B, g)
    g_d = G.T @ (sh @ g)                       # sum_k dB_k/dd * (sh[k] . g)
    return g_sh, (g_d - d * (d @ g_d)) / ln


def inria_rotation(q):
    """The matrix upstream builds from the RAW quaternion (real part first), math rows / columns."""
    r, x, y, z = np.asarray(q, np.float64)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                     [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                     [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])


def inria_cov3d(scale, rot, mod=1.0):
    m = inria_rotation(rot) @ np.diag(mod * np.asarray(scale, np.float64)[:3])
    s = m @ m.T
    return np.array([s[0, 0], s[0, 1], s[0, 2], s[1, 1], s[1, 2], s[2, 2]])


def inria_cov3d_backward(scale, rot, mod, g6):
    """dL/dscale[3], dL/drot[4] (raw quaternion: no normalisation in the forward, none here)."""
    r, x, y, z = np.asarray(rot, np.float64)
    R = inria_rotation(rot)
    s = mod * np.asarray(scale, np.float64)[:3]
    M = R @ np.diag(s)
    gS = np.array([[g6[0], 0.5 * g6[1], 0.5 * g6[2]], [0.5 * g6[1], g6[3], 0.5 * g6[4]], [0.5 * g6[2], 0.5 * g6[4], g6[5]]])
    gM = 2.0 * gS @ M
    g_scale = mod * (R * gM).sum(0)
    gR = gM * s[None, :]
    dR = {"r": np.array([[0, -2 * z, 2 * y], [2 * z, 0, -2 * x], [-2 * y, 2 * x, 0]]),
          "x": np.array([[0, 2 * y, 2 * z], [2 * y, -4 * x, -2 * r], [2 * z, 2 * r, -4 * x]]),
          "y": np.array([[-4 * y, 2 * x, 2 * r], [2 * x, 0, 2 * z], [-2 * r, 2 * z, -4 * y]]),
          "z": np.array([[-4 * z, -2 * r, 2 * x], [2 * r, -4 * z, 2 * y], [2 * x, 2 * y, 0]])}
    return g_scale, np.array([(gR * dR[k]).sum() for k in "rxyz"])


def _inria_jw(mean3, view, fx, fy, tan_fovx, tan_fovy):
    v = np.asarray(view, np.float64)
    t = np.array([v[0] * mean3[0] + v[4] * mean3[1] + v[8] * mean3[2] + v[12],
                  v[1] * mean3[0] + v[5] * mean3[1] + v[9] * mean3[2] + v[13],
                  v[2] * mean3[0] + v[6] * mean3[1] + v[10] * mean3[2] + v[14]])
    limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
    rx, ry = t[0] / t[2], t[1] / t[2]
    cx, cy = min(limx, max(-limx, rx)), min(limy, max(-limy, ry))
    tx, ty, tz = cx * t[2], cy * t[2], t[2]
    J = np.array([[fx / tz, 0.0, -fx * tx / (tz * tz)], [0.0, fy / tz, -fy * ty / (tz * tz)]])
    W = np.array([[v[0], v[4], v[8]], [v[1], v[5], v[9]], [v[2], v[6], v[10]]])
    return J, W, (tx, ty, tz), (rx != cx, ry != cy, cx, cy)


def inria_cov2d_conic(c3, mean3, view, fx, fy, tan_fovx, tan_fovy):
    J, W, _, _ = _inria_jw(mean3, view, fx, fy, tan_fovx, tan_fovy)
    P = J @ W
    cov = P @ _sigma(c3) @ P.T
    a, b, c = cov[0, 0] + 0.3, cov[0, 1], cov[1, 1] + 0.3
    det = a * c - b * b
    return np.array([c / det, -b / det, a / det])


def inria_conic_backward(c3, mean3, view, fx, fy, tan_fovx, tan_fovy, dL_dconic, dL_dcov2D=None):
    """(dL/d(6 covariance numbers), dL/dmean3 through the Jacobian's dependence on the view-space position)."""
    J, W, (tx, ty, tz), (clx, cly, cx, cy) = _inria_jw(mean3, view, fx, fy, tan_fovx, tan_fovy)
    P = J @ W
    Mw = W @ _sigma(c3) @ W.T
    cov = J @ Mw @ J.T
    a, b, c = cov[0, 0] + 0.3, cov[0, 1], cov[1, 1] + 0.3
    det = a * c - b * b
    K = np.array([[c, -b], [-b, a]]) / det
    gcov = _gcov(K, dL_dconic, dL_dcov2D)
    gS = P.T @ gcov @ P
    g6 = np.array([gS[0, 0], 2.0 * gS[0, 1], 2.0 * gS[0, 2], gS[1, 1], 2.0 * gS[1, 2], gS[2, 2]])
    gJ = 2.0 * gcov @ J @ Mw
    g_tx = gJ[0, 2] * (-fx / (tz * tz))
    g_ty = gJ[1, 2] * (-fy / (tz * tz))
    g_tz = (-gJ[0, 0] * fx - gJ[1, 1] * fy) / (tz * tz) + (gJ[0, 2] * fx * tx + gJ[1, 2] * fy * ty) * (2.0 / tz ** 3)
    gt = np.array([0.0 if clx else g_tx, 0.0 if cly else g_ty,
                   g_tz + (g_tx * cx if clx else 0.0) + (g_ty * cy if cly else 0.0)])
    return g6, W.T @ gt


def inria_project_mean2d(mean3, proj, width, height):
    p = np.asarray(proj, np.float64)
    hx = p[0] * mean3[0] + p[4] * mean3[1] + p[8] * mean3[2] + p[12]
    hy = p[1] * mean3[0] + p[5] * mean3[1] + p[9] * mean3[2] + p[13]
    hw = p[3] * mean3[0] + p[7] * mean3[1] + p[11] * mean3[2] + p[15]
    ow = 1.0 / (hw + 1e-7)
    return np.array([((hx * ow + 1.0) * width - 1.0) * 0.5, ((hy * ow + 1.0) * height - 1.0) * 0.5])


def inria_project_mean2d_backward(mean3, proj, width, height, g2):
    p = np.asarray(proj, np.float64)
    hx = p[0] * mean3[0] + p[4] * mean3[1] + p[8] * mean3[2] + p[12]
    hy = p[1] * mean3[0] + p[5] * mean3[1] + p[9] * mean3[2] + p[13]
    wp = 1e-7 + (p[3] * mean3[0] + p[7] * mean3[1] + p[11] * mean3[2] + p[15])
    g = np.zeros(3)
    for j in range(3):
        dx = 0.5 * width * (p[4 * j + 0] / wp - hx * p[4 * j + 3] / (wp * wp))
        dy = 0.5 * height * (p[4 * j + 1] / wp - hy * p[4 * j + 3] / (wp * wp))
        g[j] = dx * g2[0] + dy * g2[1]
    return g
