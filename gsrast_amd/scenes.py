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
