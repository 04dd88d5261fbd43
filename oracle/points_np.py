"""Point-splat path restated in numpy (reference apps/gsrast/gscuda/GSCuda.cu:47-108 projectPoints,
:110-155 forwardPoints). TEST INFRASTRUCTURE: imported by tests/ only.

The reference's depth test is an atomicMin followed by an unordered colour write, so with several
points on one pixel its image is timing dependent; this restatement (like the HIP path) takes the
nearest point and, among equal depths, the lowest index — one of the reference's possible outcomes.
Parity unpinned by the reference (no tests, never called)."""
import numpy as np

F = np.float32


def forward_points(means3, shs, proj, width, height, background):
    m = np.asarray(means3, F)
    p = np.asarray(proj, F)
    x, y, z, one = m[:, 0], m[:, 1], m[:, 2], F(1.0)
    hx = (p[0] * x + p[4] * y) + (p[8] * z + p[12] * one)
    hy = (p[1] * x + p[5] * y) + (p[9] * z + p[13] * one)
    hz = (p[2] * x + p[6] * y) + (p[10] * z + p[14] * one)
    hw = (p[3] * x + p[7] * y) + (p[11] * z + p[15] * one)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = F(1.0) / (hw + F(0.001))
        px, py, pz = hx * inv, hy * inv, hz * inv
        keep = ~((pz < 0) | (pz > 1) | (px < -1) | (px > 1) | (py < -1) | (py > 1)) & np.isfinite(px) & np.isfinite(py) & np.isfinite(pz)
        # (0.5f + 0.5 * x) * width is double arithmetic in the reference (the 0.5 literal), narrowed to float
        ix = ((0.5 + 0.5 * px.astype(np.float64)) * width).astype(F)
        iy = ((0.5 + 0.5 * py.astype(np.float64)) * height).astype(F)
    # round(): half away from zero
    cx = np.where(keep, np.sign(ix) * np.floor(np.abs(ix.astype(np.float64)) + 0.5), -1).astype(np.int64)
    cy = np.where(keep, np.sign(iy) * np.floor(np.abs(iy.astype(np.float64)) + 0.5), -1).astype(np.int64)
    keep &= (cx >= 0) & (cx < width) & (cy >= 0) & (cy < height)
    idx = np.nonzero(keep)[0]
    key = (pz[idx].view(np.uint32).astype(np.uint64) << np.uint64(32)) | idx.astype(np.uint64)
    pix = cy[idx] * width + cx[idx]
    winner = np.full(width * height, np.uint64(0xFFFFFFFFFFFFFFFF))
    np.minimum.at(winner, pix, key)
    out = np.empty((3, height * width), F)
    out[:] = np.asarray(background, F)[:, None]
    depth = np.ones(height * width, F)
    hit = winner != np.uint64(0xFFFFFFFFFFFFFFFF)
    w_idx = (winner[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    sh = np.asarray(shs, F)
    for c in range(3):
        out[c, hit] = F(0.4) * sh[w_idx, c] + F(0.5)
    depth[hit] = (winner[hit] >> np.uint64(32)).astype(np.uint32).view(F)
    return out.reshape(3, height, width), depth.reshape(height, width)
