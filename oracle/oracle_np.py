"""Second, independent restatement of the reference forward pass in numpy float32.

TEST INFRASTRUCTURE (same rules as gsr_oracle.cpp). Written from SURVEY.md §8(a)
(rows a3-a11, citing apps/gsrast/gscuda/GSCuda.cu), vectorised over Gaussians and over
the pixels of a tile, so it shares no code or loop structure with the scalar C++
oracle. Its job is to cross-check that oracle (integers exactly, floats to 1e-6) and
to mint the golden fixtures under tests/golden/. Parity is unpinned by the reference.
"""
from __future__ import annotations

import numpy as np

F = np.float32
TILE = 16
BATCH = 256


def _m4v(m, v):
    """glm mat4 (16 floats, column-major) times an [n,4] array of vec4."""
    m = m.astype(F)
    out = np.empty_like(v)
    for r in range(4):
        out[:, r] = (m[r] * v[:, 0] + m[4 + r] * v[:, 1]) + (m[8 + r] * v[:, 2] + m[12 + r] * v[:, 3])
    return out


def _mm3(a, b):
    """Batched glm mat3 product; matrices are dicts (c, r) -> [n] arrays."""
    return {(c, r): a[(0, r)] * b[(c, 0)] + a[(1, r)] * b[(c, 1)] + a[(2, r)] * b[(c, 2)]
            for c in range(3) for r in range(3)}


def _tr3(a):
    return {(c, r): a[(r, c)] for c in range(3) for r in range(3)}


def _f2i(x):
    """float32 -> int32 truncation, saturating, NaN -> 0 (device conversion)."""
    x64 = np.nan_to_num(x.astype(np.float64), nan=0.0, posinf=2147483647.0, neginf=-2147483648.0)
    return np.clip(np.trunc(x64), -2147483648.0, 2147483647.0).astype(np.int64).astype(np.int32)


def higher_msb(n: int) -> int:
    """GSCuda.cu:481-502."""
    msb, step = 16, 16
    while step > 1:
        step //= 2
        msb = msb + step if (n >> msb) else msb - step
    if n >> msb:
        msb += 1
    return msb


def _rect(px, py, ex, ey, gx, gy):
    """GSCuda.cu:237-259."""
    sixteen, one = F(16.0), F(1.0)
    exf, eyf = ex.astype(F), ey.astype(F)
    x0 = np.clip(_f2i((px - exf) / sixteen), 0, gx)
    y0 = np.clip(_f2i((py - eyf) / sixteen), 0, gy)
    x1 = np.clip(_f2i((((px + exf) + sixteen) - one) / sixteen), 0, gx)
    y1 = np.clip(_f2i((((py + eyf) + sixteen) - one) / sixteen), 0, gy)
    return x0.astype(np.uint32), y0.astype(np.uint32), x1.astype(np.uint32), y1.astype(np.uint32)


def preprocess(scene, cam, scale_modifier=1.0, use_rects=True):
    """Rows a3-a7 (GSCuda.cu:261-375, :771)."""
    with np.errstate(all="ignore"):
        means = scene["means3D"].astype(F)
        n = means.shape[0]
        W, H = cam.width, cam.height
        gx, gy = (W + 15) // 16, (H + 15) // 16
        view, proj = cam.view.astype(F), cam.proj.astype(F)
        tfx, tfy = F(cam.tan_fovx), F(cam.tan_fovy)
        focal = F(H) / (F(2.0) * tfy)

        ph = _m4v(proj, means)
        oow = F(1.0) / (F(0.001) + ph[:, 3])
        prx, pry, prz = oow * ph[:, 0], oow * ph[:, 1], oow * ph[:, 2]
        in_frustum = ~((prz < 0) | (prz > 1) | (prx < F(-1.3)) | (prx > F(1.3)) | (pry < F(-1.3)) | (pry > F(1.3)))

        # a4: covariance in 3D
        rot = scene["rotations"].astype(F)
        d = (rot[:, 0] * rot[:, 0] + rot[:, 1] * rot[:, 1]) + (rot[:, 2] * rot[:, 2] + rot[:, 3] * rot[:, 3])
        inv = F(1.0) / np.sqrt(d)
        x, y, z, w = (rot[:, i] * inv for i in range(4))
        dbl = lambda s, diag: (2.0 * s.astype(np.float64) - (1.0 if diag else 0.0)).astype(F)
        R = {(0, 0): dbl(x * x + y * y, True), (0, 1): dbl(y * z + x * w, False), (0, 2): dbl(y * w - x * z, False),
             (1, 0): dbl(y * z - x * w, False), (1, 1): dbl(x * x + z * z, True), (1, 2): dbl(z * w + x * y, False),
             (2, 0): dbl(y * w + x * z, False), (2, 1): dbl(z * w - x * y, False), (2, 2): dbl(x * x + w * w, True)}
        sc = scene["scales"].astype(F)
        zero = np.zeros(n, F)
        S = {(c, r): (F(scale_modifier) * sc[:, c] if c == r else zero) for c in range(3) for r in range(3)}
        RS = _mm3(R, S)
        sig = _mm3(RS, _tr3(RS))
        cov3d = np.stack([sig[(0, 0)], sig[(1, 0)], sig[(2, 0)], sig[(1, 1)], sig[(2, 1)], sig[(2, 2)]], axis=1)

        # a5: covariance in 2D
        mv = means.copy()
        mv[:, 3] = F(1.0)
        t = _m4v(view, mv)
        limx, limy = F(1.3) * tfx, F(1.3) * tfy
        tz = t[:, 2]
        tx = np.minimum(limx, np.maximum(-limx, t[:, 0] / tz)) * tz
        ty = np.minimum(limy, np.maximum(-limy, t[:, 1] / tz)) * tz
        J = {(c, r): zero for c in range(3) for r in range(3)}
        J[(0, 0)] = focal / tz
        J[(0, 2)] = (-focal * tx) / (tz * tz)
        J[(1, 1)] = focal / tz
        J[(1, 2)] = (-focal * ty) / (tz * tz)
        Wm = {(c, r): np.full(n, view[4 * r + c], F) for c in range(3) for r in range(3)}
        T = _mm3(Wm, J)
        V = {(0, 0): cov3d[:, 0], (0, 1): cov3d[:, 1], (0, 2): cov3d[:, 2],
             (1, 0): cov3d[:, 1], (1, 1): cov3d[:, 3], (1, 2): cov3d[:, 4],
             (2, 0): cov3d[:, 2], (2, 1): cov3d[:, 4], (2, 2): cov3d[:, 5]}
        C = _mm3(_mm3(_tr3(T), V), T)
        ca, cb, cc = C[(0, 0)] + F(0.3), C[(0, 1)], C[(1, 1)] + F(0.3)

        det = ca * cc - cb * cb
        has_det = in_frustum & (det != 0)
        det_inv = F(1.0) / det
        conic = np.stack([cc * det_inv, -cb * det_inv, ca * det_inv], axis=1)
        mid = F(0.5) * (ca + cc)
        root = np.sqrt(np.maximum(F(0.1), mid * mid - det))
        radius = np.ceil(F(3.0) * np.sqrt(np.maximum(mid + root, mid - root)))
        pix = (prx * F(0.5) + F(0.5)) * F(W)
        piy = (pry * F(0.5) + F(0.5)) * F(H)
        if use_rects:
            ex = _f2i(np.ceil(F(3.0) * np.sqrt(ca)))
            ey = _f2i(np.ceil(F(3.0) * cc))
        else:
            ex = ey = _f2i(radius)
        x0, y0, x1, y1 = _rect(pix, piy, ex, ey, gx, gy)
        area = (x1 - x0) * (y1 - y0)
        vis = has_det & (area != 0)

        o = {
            "radii": np.where(vis, _f2i(radius), 0).astype(np.int32),
            "tilesTouched": np.where(vis, area, 0).astype(np.uint32),
            "cov3D": np.where(in_frustum[:, None], cov3d, F(0)),
            "rects": (np.where(has_det[:, None], np.stack([ex, ey], axis=1), 0).astype(np.int32)
                      if use_rects else None),
            "rgb": np.where(vis[:, None], F(0.5) + F(0.4) * scene["shs"][:, :3].astype(F), F(0)),
            "depths": np.where(vis, prz, F(0)),
            "means2D": np.where(vis[:, None], np.stack([pix, piy], axis=1), F(0)),
            "conicOpacity": np.where(vis[:, None], np.concatenate(
                [conic, scene["opacities"].astype(F)[:, None]], axis=1), F(0)),
        }
        o["pointOffsets"] = np.cumsum(o["tilesTouched"].astype(np.uint64)).astype(np.uint32)
        o["num_rendered"] = int(o["pointOffsets"][-1]) if n else 0
        o["_rect"] = (x0, y0, x1, y1)
        return o


def bin_and_sort(o, cam):
    """Rows a8-a10 (GSCuda.cu:422-475, :794-797, :504-538)."""
    W, H = cam.width, cam.height
    gx, gy = (W + 15) // 16, (H + 15) // 16
    x0, y0, x1, y1 = o["_rect"]
    vis = o["radii"] > 0
    ids = np.nonzero(vis)[0]
    cw = (x1 - x0)[ids].astype(np.int64)
    cnt = o["tilesTouched"][ids].astype(np.int64)
    rep = np.repeat(np.arange(ids.size), cnt)
    start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    local = np.arange(rep.size) - start[rep]
    ty = y0[ids][rep].astype(np.int64) + local // cw[rep]
    tx = x0[ids][rep].astype(np.int64) + local % cw[rep]
    tile = (ty * gx + tx).astype(np.uint64)
    dbits = o["depths"].view(np.uint32)[ids][rep].astype(np.uint64)
    keys_u = (tile << np.uint64(32)) | dbits
    vals_u = ids[rep].astype(np.uint32)
    bits = 32 + higher_msb(gx * gy)
    mask = np.uint64((1 << bits) - 1)
    order = np.argsort(keys_u & mask, kind="stable")
    keys, vals = keys_u[order], vals_u[order]
    R = keys.size
    ranges = np.zeros((gx * gy, 2), np.uint32)
    if R > 0:
        tl = (keys >> np.uint64(32)).astype(np.int64)
        ranges[tl[0], 0] = 0
        chg = np.nonzero(tl[1:] != tl[:-1])[0] + 1
        ranges[tl[chg - 1], 1] = chg
        ranges[tl[chg], 0] = chg
        if R > 1:                                   # the idx==R-1 close sits in the else branch (:533)
            ranges[tl[-1], 1] = R
    o.update(keys_unsorted=keys_u, values_unsorted=vals_u, keys=keys, values=vals, ranges=ranges)
    return o


def blend(o, cam, background=(0.0, 0.0, 0.0), out_init=None, t_cutoff=0.001):
    """Row a11 (GSCuda.cu:543-677): per tile, all 256 pixels advance together through
    the sorted list, one record at a time, with boolean masks for skip/done."""
    W, H = cam.width, cam.height
    gx, gy = (W + 15) // 16, (H + 15) // 16
    bg = np.asarray(background, dtype=F)
    out = np.zeros((3, H, W), F) if out_init is None else np.array(out_init, dtype=F)
    finalT = np.zeros((H, W), F)
    ncontrib = np.zeros((H, W), np.uint32)
    staged = 0
    if o["num_rendered"] == 0:
        return dict(out_color=out, finalT=finalT, nContrib=ncontrib, records_staged=0)
    ly, lx = np.divmod(np.arange(256), 16)
    with np.errstate(all="ignore"):
        for tyi in range(gy):
            for txi in range(gx):
                r0, r1 = (int(v) for v in o["ranges"][tyi * gx + txi])
                px, py = txi * 16 + lx, tyi * 16 + ly
                inside = (px < min(txi * 16 + 16, W)) & (py < min(tyi * 16 + 16, H))
                done = ~inside
                T = np.ones(256, F)
                col = np.zeros((256, 3), F)
                contrib = np.zeros(256, np.uint32)
                last = np.zeros(256, np.uint32)
                fx, fy = px.astype(F), py.astype(F)
                rounds = ((r1 - r0) % (1 << 32) + 255) // 256
                for i in range(rounds):
                    if done.all():
                        break
                    lo = r0 + i * 256
                    hi = min(lo + 256, r1)
                    staged += hi - lo
                    for k in range(lo, hi):
                        if done.all():
                            break           # no pixel left; skipping the rest changes nothing observable
                        g = int(o["values"][k])
                        act = ~done
                        contrib[act] += 1
                        gxy, con, rgb = o["means2D"][g], o["conicOpacity"][g], o["rgb"][g]
                        dx, dy = gxy[0] - fx, gxy[1] - fy
                        power = F(-0.5) * (con[0] * dx * dx + con[2] * dy * dy) - con[1] * dx * dy
                        alpha = np.minimum(F(0.99), con[3] * np.exp(power))
                        test = T * (F(1.0) - alpha)
                        live = act & ~(power > 0) & ~(alpha < F(1.0) / F(255.0))
                        stop = live & (test < F(t_cutoff))
                        upd = live & ~stop
                        done |= stop
                        for c in range(3):
                            col[upd, c] = col[upd, c] + rgb[c] * alpha[upd] * T[upd]
                        T[upd] = test[upd]
                        last[upd] = contrib[upd]
                ys, xs = py[inside], px[inside]
                finalT[ys, xs] = T[inside]
                ncontrib[ys, xs] = last[inside]
                for c in range(3):
                    out[c, ys, xs] = col[inside, c] + T[inside] * bg[c]
    return dict(out_color=out, finalT=finalT, nContrib=ncontrib, records_staged=staged)


def forward(scene, cam, background=(0.0, 0.0, 0.0), use_rects=True, scale_modifier=1.0, out_init=None):
    o = preprocess(scene, cam, scale_modifier, use_rects)
    if o["num_rendered"] > 0:
        bin_and_sort(o, cam)
    else:
        gx, gy = (cam.width + 15) // 16, (cam.height + 15) // 16
        o.update(keys_unsorted=np.zeros(0, np.uint64), values_unsorted=np.zeros(0, np.uint32),
                 keys=np.zeros(0, np.uint64), values=np.zeros(0, np.uint32),
                 ranges=np.zeros((gx * gy, 2), np.uint32))
    o.update(blend(o, cam, background, out_init))
    o.pop("_rect", None)
    return o
