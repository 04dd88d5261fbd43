"""Synthetic splat scenes for tests and benchmarks.

Every scene is defined arithmetically from splitmix64 so any host reproduces the same
bits: uniform u_k = (splitmix64(seed, k) >> 40) * 2**-24. A scene is a dict of float32
arrays laid out as the reference uploads them (apps/gsrast/GSGaussians.cpp:109-137,
SplatData.cpp:50-54): means3D vec4 (w = 1), scales vec4 (already exp-activated,
w = e), rotations vec4 (real part first, unit), opacities (sigmoid-activated),
shs 48 floats per splat in raw PLY order (DC first).
"""
from __future__ import annotations

import math

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """Outputs start .. start+count-1 of the splitmix64 stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


class Stream:
    """Sequential reader over one splitmix64 stream."""

    def __init__(self, seed: int):
        self.seed = seed
        self.pos = 0

    def uniform(self, *shape: int) -> np.ndarray:
        n = int(np.prod(shape))
        z = splitmix64(self.seed, self.pos, n)
        self.pos += n
        return ((z >> np.uint64(40)).astype(np.float64) * (2.0 ** -24)).reshape(shape)

    def normal(self, *shape: int) -> np.ndarray:
        u1 = self.uniform(*shape)
        u2 = self.uniform(*shape)
        return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def _sigmoid(x: np.ndarray) -> np.ndarray:
    return 1.0 / (1.0 + np.exp(-x))


def _pack(pos, scale, quat, opacity, dc, sh_rest=None) -> dict:
    n = pos.shape[0]
    means = np.ones((n, 4), dtype=np.float32)
    means[:, :3] = pos
    scales = np.full((n, 4), math.e, dtype=np.float32)      # exp(vec4(s, 1)).w, SplatData.cpp:51
    scales[:, :3] = scale
    shs = np.zeros((n, 48), dtype=np.float32)
    shs[:, :3] = dc
    if sh_rest is not None:
        shs[:, 3:] = sh_rest
    return {
        "means3D": means,
        "scales": scales,
        "rotations": np.ascontiguousarray(quat, dtype=np.float32),
        "opacities": np.ascontiguousarray(opacity, dtype=np.float32),
        "shs": shs,
    }


def isotropic_scene(n: int = 1000, seed: int = 42) -> dict:
    """BASELINE config 1: n isotropic Gaussians in [-1,1]^3."""
    s = Stream(seed)
    pos = s.uniform(n, 3) * 2.0 - 1.0
    sc = np.exp(math.log(0.02) + s.uniform(n, 1) * (math.log(0.2) - math.log(0.02)))
    quat = np.zeros((n, 4))
    quat[:, 0] = 1.0
    opacity = _sigmoid(-2.0 + s.uniform(n) * 6.0)
    dc = s.uniform(n, 3) * 3.0 - 1.5
    return _pack(pos, np.repeat(sc, 3, axis=1), quat, opacity, dc)


def _unit_quats(s: Stream, n: int) -> np.ndarray:
    q = s.normal(n, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def garden_like_scene(n: int = 5_834_784, seed: int = 43) -> dict:
    """Synthetic stand-in for BASELINE config 2/3 (the Mip-NeRF360 garden .ply is not
    available offline): same splat count, anisotropic, log-normal scales."""
    s = Stream(seed)
    pos = s.normal(n, 3) * np.array([4.0, 1.5, 4.0])
    scale = np.exp(-4.5 + 1.2 * s.normal(n, 3))
    quat = _unit_quats(s, n)
    opacity = _sigmoid(3.0 * s.normal(n))
    dc = s.normal(n, 3)
    return _pack(pos, scale, quat, opacity, dc)


def stress_scene(n: int = 50_000_000, seed: int = 44, full_sh: bool = False) -> dict:
    """BASELINE config 4: anisotropic Gaussians in [-10,10]^3. The gscuda semantics read
    only the DC term, so the 45 higher-order floats are generated only on request."""
    s = Stream(seed)
    pos = s.uniform(n, 3) * 20.0 - 10.0
    scale = np.exp(math.log(0.005) + s.uniform(n, 3) * (math.log(0.05) - math.log(0.005)))
    quat = _unit_quats(s, n)
    opacity = _sigmoid(-2.0 + s.uniform(n) * 6.0)
    dc = 0.3 * s.normal(n, 3)
    rest = 0.3 * s.normal(n, 45) if full_sh else None
    return _pack(pos, scale, quat, opacity, dc, rest)


def _quat_from_columns(t1: np.ndarray, t2: np.ndarray, t3: np.ndarray) -> np.ndarray:
    """Unit quaternions (real part first) of the rotations whose matrix has columns t1, t2, t3 (right-handed, orthonormal)."""
    m00, m10, m20 = t1[:, 0], t1[:, 1], t1[:, 2]
    m01, m11, m21 = t2[:, 0], t2[:, 1], t2[:, 2]
    m02, m12, m22 = t3[:, 0], t3[:, 1], t3[:, 2]
    # the largest of the four squared components is computed from the trace, the others from the off-diagonal sums
    q = np.empty((t1.shape[0], 4))
    w2, x2 = 1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22
    y2, z2 = 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22
    pick = np.argmax(np.stack([w2, x2, y2, z2], axis=1), axis=1)
    for k, (d2, comps) in enumerate(((w2, (None, m21 - m12, m02 - m20, m10 - m01)),
                                     (x2, (m21 - m12, None, m01 + m10, m02 + m20)),
                                     (y2, (m02 - m20, m01 + m10, None, m12 + m21)),
                                     (z2, (m10 - m01, m02 + m20, m12 + m21, None)))):
        sel = pick == k
        if not sel.any():
            continue
        r = np.sqrt(np.maximum(d2[sel], 1e-12))
        for c in range(4):
            q[sel, c] = 0.5 * r if comps[c] is None else 0.5 * comps[c][sel] / r
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def trained_like(n: int = 1_000_000, seed: int = 45) -> dict:
    """What a TRAINED scene looks like to the rasterizer, as far as a generator can say (the garden .ply itself is not
    available offline; SplatData.cpp:114-156 loads such files): splats lie ON surfaces — a ground plane, a few hundred
    plane patches and shells — and are FLAT (the scale along the surface normal 10-100 times smaller than the two in the
    surface), their in-surface scales are log-normal with a heavy tail, opacity is bimodal (most splats nearly opaque, a
    third nearly transparent), and some 500 huge splats far out stand for the background. Deterministic (splitmix64)."""
    s = Stream(seed)
    n_bg = min(500, n // 20)
    n_fg = n - n_bg
    n_surf = 320
    kind = s.uniform(n_surf)                                   # < 0.7: plane patch, else shell
    centre = s.normal(n_surf, 3) * np.array([4.0, 1.2, 4.0])
    normal = s.normal(n_surf, 3)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    ext = np.exp(math.log(0.3) + s.uniform(n_surf, 2) * (math.log(3.0) - math.log(0.3)))      # plane half-extents
    radius = np.exp(math.log(0.2) + s.uniform(n_surf) * (math.log(1.5) - math.log(0.2)))
    base_dc = s.normal(n_surf, 3)
    is_plane = kind < 0.7
    # surface 0 is the ground: a big plane below the camera (the inverted-up convention: +y is down)
    centre[0], normal[0], ext[0], is_plane[0] = (0.0, 1.6, 0.0), (0.0, -1.0, 0.0), (12.0, 12.0), True
    area = np.where(is_plane, 4.0 * ext[:, 0] * ext[:, 1], 4.0 * math.pi * radius ** 2)
    area[0] *= 0.35                                            # (the ground is sampled more thinly than the objects)
    cum = np.cumsum(area) / area.sum()
    surf = np.minimum(np.searchsorted(cum, s.uniform(n_fg)), n_surf - 1)
    # a frame (t1, t2, nrm) per splat: the surface's for planes, the radial one for shells
    nrm = normal[surf]
    uv = s.uniform(n_fg, 2) * 2.0 - 1.0
    d = s.normal(n_fg, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    shell = ~is_plane[surf]
    nrm = np.where(shell[:, None], d, nrm)
    helper = np.where(np.abs(nrm[:, 1:2]) < 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t1 = np.cross(helper, nrm)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    pos = np.where(shell[:, None], centre[surf] + radius[surf][:, None] * d,
                   centre[surf] + (uv[:, 0] * ext[surf, 0])[:, None] * t1 + (uv[:, 1] * ext[surf, 1])[:, None] * t2)
    pos += 0.01 * s.normal(n_fg, 3)                            # (trained surfaces are not perfectly thin)
    # in-surface rotation, then scales: two log-normal in-surface ones (2 % of them 4.5-20 times larger), a thin one
    phi = s.uniform(n_fg) * 2.0 * math.pi
    a1 = np.cos(phi)[:, None] * t1 + np.sin(phi)[:, None] * t2
    a2 = np.cross(nrm, a1)
    quat = _quat_from_columns(a1, a2, nrm)
    st = np.exp(-4.6 + 0.9 * s.normal(n_fg, 2))
    st *= np.where(s.uniform(n_fg) < 0.02, np.exp(1.5 + 1.5 * s.uniform(n_fg)), 1.0)[:, None]
    sn = st.min(axis=1) * np.exp(-(math.log(10.0) + s.uniform(n_fg) * math.log(10.0)))
    scale = np.concatenate([st, sn[:, None]], axis=1)
    opacity = np.where(s.uniform(n_fg) < 0.65, _sigmoid(4.0 + s.normal(n_fg)), _sigmoid(-2.5 + s.normal(n_fg)))
    dc = base_dc[surf] + 0.25 * s.normal(n_fg, 3)
    # the background: huge flat splats on a far shell, facing the origin
    bd = s.normal(n_bg, 3)
    bd /= np.linalg.norm(bd, axis=1, keepdims=True)
    bpos = bd * (25.0 + 15.0 * s.uniform(n_bg))[:, None]
    bh = np.where(np.abs(bd[:, 1:2]) < 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    b1 = np.cross(bh, bd)
    b1 /= np.linalg.norm(b1, axis=1, keepdims=True)
    bquat = _quat_from_columns(b1, np.cross(bd, b1), bd)
    bst = np.exp(math.log(2.0) + s.uniform(n_bg, 2) * (math.log(8.0) - math.log(2.0)))
    bscale = np.concatenate([bst, 0.05 * bst.min(axis=1, keepdims=True)], axis=1)
    bop = _sigmoid(3.0 + s.normal(n_bg))
    bdc = 0.5 * s.normal(n_bg, 3)
    # the background splats are dealt through the file as a trainer's densification leaves them: anywhere
    order = np.argsort(splitmix64(seed ^ 0x5bd1e995, 0, n), kind="stable")
    cat = lambda a, b: np.concatenate([a, b], axis=0)[order]
    return _pack(cat(pos, bpos), cat(scale, bscale), cat(quat, bquat), cat(opacity, bop), cat(dc, bdc))


# ---- the same scenes generated on a torch device -----------------------------------------------------
# 50 M splats x 48 SH floats are 9.6 GB: building them with numpy on the host and uploading takes minutes.
# The arithmetic definition is the same (splitmix64 -> 24-bit uniforms -> float64 transforms -> float32);
# the float64 exp / log / cos of the device may differ from numpy's in the last place, which survives the
# rounding to float32 about once in 2^29 values, so a test that compares against the CPU oracle takes its
# sample FROM the device scene (scene_rows) instead of regenerating it.
def _to_i64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def splitmix64_torch(seed: int, start: int, count: int, device):
    """Outputs start .. start+count-1 of the stream as int64 bit patterns (two's-complement wrap = uint64 arithmetic)."""
    import torch

    def lsr(z, s):      # logical shift right on int64
        return (z >> s) & ((1 << (64 - s)) - 1)
    k = torch.arange(start + 1, start + count + 1, dtype=torch.int64, device=device)
    z = k * _to_i64(int(_GOLDEN)) + _to_i64(seed)
    z = (z ^ lsr(z, 30)) * _to_i64(int(_M1))
    z = (z ^ lsr(z, 27)) * _to_i64(int(_M2))
    return z ^ lsr(z, 31)


class DeviceStream:
    """Stream with random access: uniform_at(offset, count) are outputs pos+offset .. of the numpy Stream."""

    def __init__(self, seed: int, device):
        self.seed, self.device, self.pos = seed, device, 0

    def uniform_at(self, start: int, count: int):
        import torch
        z = splitmix64_torch(self.seed, start, count, self.device)
        return ((z >> 40) & ((1 << 24) - 1)).to(torch.float64) * (2.0 ** -24)

    def uniform(self, *shape: int):
        n = int(np.prod(shape))
        u = self.uniform_at(self.pos, n).reshape(shape)
        self.pos += n
        return u

    def normal(self, *shape: int):
        import torch
        u1 = self.uniform(*shape)
        u2 = self.uniform(*shape)
        return torch.sqrt(-2.0 * torch.log(1.0 - u1)) * torch.cos(2.0 * math.pi * u2)

    def normal_rows(self, rows: int, cols: int, r0: int, r1: int):
        """Rows [r0, r1) of what normal(rows, cols) returns, without advancing (call skip afterwards)."""
        import torch
        u1 = self.uniform_at(self.pos + r0 * cols, (r1 - r0) * cols)
        u2 = self.uniform_at(self.pos + rows * cols + r0 * cols, (r1 - r0) * cols)
        return (torch.sqrt(-2.0 * torch.log(1.0 - u1)) * torch.cos(2.0 * math.pi * u2)).reshape(r1 - r0, cols)

    def skip(self, count: int):
        self.pos += count


def _pack_device(pos, scale, quat, opacity, dc, device) -> dict:
    import torch
    n = pos.shape[0]
    means = torch.ones((n, 4), dtype=torch.float32, device=device)
    means[:, :3] = pos.to(torch.float32)
    scales = torch.full((n, 4), math.e, dtype=torch.float32, device=device)
    scales[:, :3] = scale.to(torch.float32)
    shs = torch.zeros((n, 48), dtype=torch.float32, device=device)
    shs[:, :3] = dc.to(torch.float32)
    return {"means3D": means, "scales": scales, "rotations": quat.to(torch.float32).contiguous(),
            "opacities": opacity.to(torch.float32).contiguous(), "shs": shs}


def stress_scene_device(n: int = 50_000_000, seed: int = 44, device="cuda:0", full_sh: bool = False,
                        sh_rows_per_chunk: int = 4_000_000) -> dict:
    """stress_scene (BASELINE config 4) as float32 device tensors."""
    import torch
    s = DeviceStream(seed, device)
    pos = s.uniform(n, 3) * 20.0 - 10.0
    scale = torch.exp(math.log(0.005) + s.uniform(n, 3) * (math.log(0.05) - math.log(0.005)))
    q = s.normal(n, 4)
    q = q / torch.linalg.norm(q, dim=1, keepdim=True)
    opacity = 1.0 / (1.0 + torch.exp(-(-2.0 + s.uniform(n) * 6.0)))
    dc = 0.3 * s.normal(n, 3)
    out = _pack_device(pos, scale, q, opacity, dc, device)
    del pos, scale, q, opacity, dc
    if full_sh:
        for r0 in range(0, n, sh_rows_per_chunk):
            r1 = min(n, r0 + sh_rows_per_chunk)
            out["shs"][r0:r1, 3:] = (0.3 * s.normal_rows(n, 45, r0, r1)).to(torch.float32)
        s.skip(2 * n * 45)
    return out


def garden_like_scene_device(n: int = 5_834_784, seed: int = 43, device="cuda:0") -> dict:
    """garden_like_scene as float32 device tensors."""
    import torch
    s = DeviceStream(seed, device)
    pos = s.normal(n, 3) * torch.tensor([4.0, 1.5, 4.0], dtype=torch.float64, device=device)
    scale = torch.exp(-4.5 + 1.2 * s.normal(n, 3))
    q = s.normal(n, 4)
    q = q / torch.linalg.norm(q, dim=1, keepdim=True)
    opacity = 1.0 / (1.0 + torch.exp(-(3.0 * s.normal(n))))
    dc = s.normal(n, 3)
    return _pack_device(pos, scale, q, opacity, dc, device)


def scene_rows(scene: dict, rows) -> dict:
    """Host (numpy float32) copy of the selected rows of a scene held as numpy arrays or torch tensors."""
    out = {}
    for k, v in scene.items():
        sub = v[rows]
        out[k] = np.ascontiguousarray(sub.cpu().numpy() if hasattr(sub, "cpu") else sub, dtype=np.float32)
    return out
