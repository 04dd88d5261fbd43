"""Host-side camera preparation: the argument prep `GSGaussians::draw` does before
it calls the rasterizer (reference: apps/gsrast/GSGaussians.cpp:155-176) and the
`FirstPersonCamera::update` matrices it starts from (FirstPersonCamera.cpp:28-38).

glm is not installable here, so `look_at` / `perspective` restate glm's default
right-handed, NDC z in [-1, 1] formulas in float32. Matrices are returned as
16 float32 in glm column-major order (element (r, c) at [4*c + r]).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

F32 = np.float32


def _normalize(v: np.ndarray) -> np.ndarray:
    v = v.astype(F32)
    return (v * (F32(1.0) / np.sqrt(np.dot(v, v), dtype=F32))).astype(F32)


def look_at(eye, center, up) -> np.ndarray:
    """glm::lookAtRH. Returns a 4x4 float32 array indexed [row, col]."""
    eye = np.asarray(eye, dtype=F32)
    f = _normalize(np.asarray(center, dtype=F32) - eye)
    s = _normalize(np.cross(f, np.asarray(up, dtype=F32)).astype(F32))
    u = np.cross(s, f).astype(F32)
    m = np.eye(4, dtype=F32)
    m[0, 0:3] = s
    m[1, 0:3] = u
    m[2, 0:3] = -f
    m[0, 3] = -np.dot(s, eye)
    m[1, 3] = -np.dot(u, eye)
    m[2, 3] = np.dot(f, eye)
    return m


def perspective(fovy: float, aspect: float, near: float, far: float) -> np.ndarray:
    """glm::perspectiveRH_NO. Returns a 4x4 float32 array indexed [row, col]."""
    t = F32(math.tan(F32(fovy) / F32(2.0)))
    m = np.zeros((4, 4), dtype=F32)
    m[0, 0] = F32(1.0) / (F32(aspect) * t)
    m[1, 1] = F32(1.0) / t
    m[2, 2] = -(F32(far) + F32(near)) / (F32(far) - F32(near))
    m[3, 2] = F32(-1.0)
    m[2, 3] = -(F32(2.0) * F32(far) * F32(near)) / (F32(far) - F32(near))
    return m


@dataclass
class Camera:
    """The 32 matrix floats + scalars one `forward` call takes."""

    view: np.ndarray        # 16 f32, column-major, row 2 negated (GSGaussians.cpp:160-169)
    proj: np.ndarray        # 16 f32, column-major, perspective * view (not negated)
    cam_pos: np.ndarray     # 3 f32 (unused by the gscuda semantics)
    tan_fovx: float
    tan_fovy: float
    width: int
    height: int


def first_person_camera(position, yaw: float, pitch: float, fov: float, near: float, far: float,
                        width: int, height: int, invert_up: bool = True) -> Camera:
    """FirstPersonCamera::update (FirstPersonCamera.cpp:28-38) followed by the
    per-frame prep of GSGaussians::draw (GSGaussians.cpp:157-176)."""
    up = np.array([0.0, -1.0 if invert_up else 1.0, 0.0], dtype=F32)
    front = np.array([math.cos(pitch) * math.sin(yaw), math.sin(pitch), math.cos(pitch) * math.cos(yaw)],
                     dtype=F32)
    pos = np.asarray(position, dtype=F32)
    view = look_at(pos, pos + front, up)
    persp = perspective(fov, F32(width) / F32(height), near, far)
    proj = (persp @ view).astype(F32)
    view = view.copy()
    view[2, :] *= F32(-1.0)
    tan_fovy = float(F32(math.tan(F32(fov) * F32(0.5))))
    tan_fovx = float(F32(tan_fovy) * (F32(width) / F32(height)))
    return Camera(view=np.ascontiguousarray(view.T).reshape(16).copy(),
                  proj=np.ascontiguousarray(proj.T).reshape(16).copy(),
                  cam_pos=pos.copy(), tan_fovx=tan_fovx, tan_fovy=tan_fovy, width=width, height=height)


def default_camera(width: int, height: int, near: float = 0.01, far: float = 100.0,
                   position=(0.0, 0.0, -5.0)) -> Camera:
    """The reference app's initial pose: eye (0,0,-5) looking at the origin, inverted up,
    fov 45 degrees (GSRastWindow.cpp:20,30-37; Config.hpp:23)."""
    return first_person_camera(position, 0.0, 0.0, math.radians(45.0), near, far, width, height, True)
