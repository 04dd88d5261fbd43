"""numpy float32 restatement of the UPSTREAM forward rasterizer semantics (graphdeco-inria/
diff-gaussian-rasterization), the `inria` profile of SURVEY.md §8(f)-2 / Divergence table D1-D12.

TEST INFRASTRUCTURE. The upstream sources are not in this container (empty submodule), so this is
written from the published algorithm: parity is UNPINNED — it checks that the HIP `inria` path and
this restatement agree, and a few closed-form facts (SH basis constants, DC colour), nothing more.

Inputs use the reference app's buffers: means3D / scales vec4-strided, rotations (real part first),
shs [N][16][3] coefficient-major (upstream layout), view / proj 16 floats column-major.
"""
from __future__ import annotations

import numpy as np

from .oracle_np import _f2i, _mm3, _tr3, bin_and_sort, blend

F = np.float32
SH_C0 = F(0.28209479177387814)
SH_C1 = F(0.4886025119029199)
SH_C2 = [F(1.0925484305920792), F(-1.0925484305920792), F(0.31539156525252005), F(-1.0925484305920792), F(0.5462742152960396)]
SH_C3 = [F(-0.5900435899266435), F(2.890611442640554), F(-0.4570457994644658), F(0.3731763325901154),
         F(-0.4570457994644658), F(1.445305721320277), F(-0.5900435899266435)]


def _p4x3(m, p):
    return np.stack([m[r] * p[:, 0] + m[4 + r] * p[:, 1] + m[8 + r] * p[:, 2] + m[12 + r] for r in range(3)], axis=1)


def _p4x4(m, p):
    return np.stack([m[r] * p[:, 0] + m[4 + r] * p[:, 1] + m[8 + r] * p[:, 2] + m[12 + r] for r in range(4)], axis=1)


def sh_to_rgb(deg, pos, campos, shs):
    d = pos - campos[None, :]
    d = d / np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])[:, None]
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    sh = shs.reshape(-1, 16, 3).astype(F)
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (F(2.0) * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (F(3.0) * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (F(4.0) * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (F(2.0) * zz - F(3.0) * xx - F(3.0) * yy) * sh[:, 12]
                       + SH_C3[4] * x * (F(4.0) * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - F(3.0) * yy) * sh[:, 15])
    res = res + F(0.5)
    return np.maximum(res, F(0.0)).astype(F), res < 0


def preprocess(scene, cam, deg=3, scale_modifier=1.0):
    with np.errstate(all="ignore"):
        means = scene["means3D"].astype(F)
        n = means.shape[0]
        W, H = cam.width, cam.height
        gx, gy = (W + 15) // 16, (H + 15) // 16
        view, proj = cam.view.astype(F), cam.proj.astype(F)
        tfx, tfy = F(cam.tan_fovx), F(cam.tan_fovy)
        fx, fy = F(W) / (F(2.0) * tfx), F(H) / (F(2.0) * tfy)
        p = means[:, :3]
        pv = _p4x3(view, p)
        vis = ~(pv[:, 2] <= F(0.2))
        ph = _p4x4(proj, p)
        pw = F(1.0) / (ph[:, 3] + F(0.0000001))
        prx, pry = ph[:, 0] * pw, ph[:, 1] * pw
        # cov3D: q not normalised; R written row-wise into a column-major constructor
        q = scene["rotations"].astype(F)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        one, two = F(1.0), F(2.0)
        R = {(0, 0): one - two * (y * y + z * z), (0, 1): two * (x * y - r * z), (0, 2): two * (x * z + r * y),
             (1, 0): two * (x * y + r * z), (1, 1): one - two * (x * x + z * z), (1, 2): two * (y * z - r * x),
             (2, 0): two * (x * z - r * y), (2, 1): two * (y * z + r * x), (2, 2): one - two * (x * x + y * y)}
        sc = scene["scales"].astype(F)
        zero = np.zeros(n, F)
        S = {(c, rr): (F(scale_modifier) * sc[:, c] if c == rr else zero) for c in range(3) for rr in range(3)}
        M = _mm3(S, R)
        sig = _mm3(_tr3(M), M)
        cov3d = np.stack([sig[(0, 0)], sig[(0, 1)], sig[(0, 2)], sig[(1, 1)], sig[(1, 2)], sig[(2, 2)]], axis=1)
        # cov2D
        limx, limy = F(1.3) * tfx, F(1.3) * tfy
        tz = pv[:, 2]
        tx = np.minimum(limx, np.maximum(-limx, pv[:, 0] / tz)) * tz
        ty = np.minimum(limy, np.maximum(-limy, pv[:, 1] / tz)) * tz
        J = {(c, rr): zero for c in range(3) for rr in range(3)}
        J[(0, 0)] = fx / tz
        J[(0, 2)] = -(fx * tx) / (tz * tz)
        J[(1, 1)] = fy / tz
        J[(1, 2)] = -(fy * ty) / (tz * tz)
        Wm = {(c, rr): np.full(n, view[4 * rr + c], F) for c in range(3) for rr in range(3)}
        T = _mm3(Wm, J)
        V = {(0, 0): cov3d[:, 0], (0, 1): cov3d[:, 1], (0, 2): cov3d[:, 2], (1, 0): cov3d[:, 1], (1, 1): cov3d[:, 3],
             (1, 2): cov3d[:, 4], (2, 0): cov3d[:, 2], (2, 1): cov3d[:, 4], (2, 2): cov3d[:, 5]}
        C = _mm3(_mm3(_tr3(T), _tr3(V)), T)
        ca, cb, cc = C[(0, 0)] + F(0.3), C[(0, 1)], C[(1, 1)] + F(0.3)
        det = ca * cc - cb * cb
        vis &= det != 0
        det_inv = F(1.0) / det
        conic = np.stack([cc * det_inv, -cb * det_inv, ca * det_inv], axis=1)
        mid = F(0.5) * (ca + cc)
        root = np.sqrt(np.maximum(F(0.1), mid * mid - det))
        radius = np.ceil(F(3.0) * np.sqrt(np.maximum(mid + root, mid - root)))
        # ndc2Pix runs in double (the 1.0 / 0.5 literals) and narrows to float
        pix = (((prx.astype(np.float64) + 1.0) * W - 1.0) * 0.5).astype(F)
        piy = (((pry.astype(np.float64) + 1.0) * H - 1.0) * 0.5).astype(F)
        ri = _f2i(radius)
        rf = ri.astype(F)
        x0 = np.clip(_f2i((pix - rf) / F(16)), 0, gx).astype(np.uint32)
        y0 = np.clip(_f2i((piy - rf) / F(16)), 0, gy).astype(np.uint32)
        x1 = np.clip(_f2i((((pix + rf) + F(16)) - F(1)) / F(16)), 0, gx).astype(np.uint32)
        y1 = np.clip(_f2i((((piy + rf) + F(16)) - F(1)) / F(16)), 0, gy).astype(np.uint32)
        area = (x1 - x0) * (y1 - y0)
        vis &= area != 0
        rgb, clamped = sh_to_rgb(deg, p, cam.cam_pos.astype(F), scene["shs"].astype(F))
        o = {
            "radii": np.where(vis, ri, 0).astype(np.int32),
            "tilesTouched": np.where(vis, area, 0).astype(np.uint32),
            "rgb": np.where(vis[:, None], rgb, F(0)),
            "depths": np.where(vis, pv[:, 2], F(0)),
            "means2D": np.where(vis[:, None], np.stack([pix, piy], axis=1), F(0)),
            "conicOpacity": np.where(vis[:, None], np.concatenate([conic, scene["opacities"].astype(F)[:, None]], axis=1), F(0)),
            "rects": None,
        }
        o["pointOffsets"] = np.cumsum(o["tilesTouched"].astype(np.uint64)).astype(np.uint32)
        o["num_rendered"] = int(o["pointOffsets"][-1]) if n else 0
        o["_rect"] = (x0, y0, x1, y1)
        return o


def forward(scene, cam, background=(0.0, 0.0, 0.0), deg=3, scale_modifier=1.0):
    o = preprocess(scene, cam, deg, scale_modifier)
    gx, gy = (cam.width + 15) // 16, (cam.height + 15) // 16
    if o["num_rendered"] > 0:
        bin_and_sort(o, cam)
        if o["num_rendered"] == 1:            # upstream closes the last tile unconditionally (D12)
            t = int(o["keys"][0] >> np.uint64(32))
            o["ranges"][t] = (0, 1)
    else:
        o.update(keys=np.zeros(0, np.uint64), values=np.zeros(0, np.uint32), ranges=np.zeros((gx * gy, 2), np.uint32))
    # upstream runs the tile loop even when nothing was binned: every pixel then gets the background
    res = blend(dict(o, num_rendered=max(1, o["num_rendered"])), cam, background, t_cutoff=1e-4)
    o.update(res)
    o.pop("_rect", None)
    return o
