"""3DGS .ply scene loading: the role of `SplatData` in the reference
(apps/gsrast/SplatData.{hpp,cpp}). Header parsing and file I/O happen on the host exactly as
the reference does them; the per-splat activations run on the GPU (`gsr_ply_activate`).

File format (SplatData.hpp:17-25, SplatData.cpp:114-156): after the header, N records of 62
little-endian float32 — position 3, normal 3, f_dc/f_rest 48, opacity 1, scale 3, rotation 4 —
in that fixed order; property names in the header are not interpreted.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi

RECORD_FLOATS = 62


def parse_header(path: str) -> tuple[int, int]:
    """(vertex count, byte offset of the first record). Raises on an unreadable header."""
    n, off = C.c_int(0), C.c_longlong(0)
    rc = _capi.lib().gsr_ply_parse_header(path.encode(), C.byref(n), C.byref(off))
    if rc != _capi.GSR_OK or off.value < 0:
        raise ValueError(f"{path}: not a readable .ply header (no end_header line)")
    return int(n.value), int(off.value)


def load_ply(path: str, device="cuda:0", sh_layout: str = "file") -> dict:
    """Loads a scene to the device in the layout `SplatRasterizer.configure_from_scene` takes.
    Also returns the bounding box and centre the reference computes (SplatData.cpp:55-62).

    sh_layout: "file" keeps the 48 SH floats as they lie in the file (f_dc, then f_rest channel-major) — what the
    reference does, and all its DC-only colour needs; "coefficient_major" transposes f_rest to [16][3], the layout
    `draw(semantics="inria", sh_degree >= 1)` reads (GSR_SH_LAYOUT_*, include/gsrast_amd.h)."""
    layout = {"file": _capi.GSR_SH_LAYOUT_FILE, "coefficient_major": _capi.GSR_SH_LAYOUT_COEFFICIENT_MAJOR}[sh_layout]
    n, off = parse_header(path)
    raw = np.fromfile(path, dtype="<f4", offset=off, count=n * RECORD_FLOATS)
    if raw.size < n * RECORD_FLOATS:                  # the reference rejects a short file (:147-152)
        raise ValueError(f"{path}: file ends before {n} records")
    dev = torch.device(device)
    raw_dev = torch.from_numpy(raw).to(dev)
    out = {
        "means3D": torch.empty((n, 4), dtype=torch.float32, device=dev),
        "scales": torch.empty((n, 4), dtype=torch.float32, device=dev),
        "rotations": torch.empty((n, 4), dtype=torch.float32, device=dev),
        "opacities": torch.empty((n,), dtype=torch.float32, device=dev),
        "shs": torch.empty((n, 48), dtype=torch.float32, device=dev),
    }
    with torch.cuda.device(dev):
        rc = _capi.lib().gsr_ply_activate_layout(raw_dev.data_ptr(), n, out["means3D"].data_ptr(), out["scales"].data_ptr(),
                                                 out["rotations"].data_ptr(), out["opacities"].data_ptr(),
                                                 out["shs"].data_ptr(), layout, torch.cuda.current_stream(dev).cuda_stream)
    _capi.check(rc, "gsr_ply_activate_layout")
    out["sh_layout"] = sh_layout
    torch.cuda.current_stream(dev).synchronize()
    pos = out["means3D"][:, :3]
    out["bbox_min"], out["bbox_max"] = pos.min(0).values, pos.max(0).values
    out["center"] = pos.mean(0)
    return out


def write_ply(path: str, position, sh, opacity_logit, log_scale, rotation, normal=None) -> None:
    """Writes a scene in the standard 3DGS property order (used by tests and tools)."""
    position = np.asarray(position, np.float32)
    n = position.shape[0]
    rec = np.zeros((n, RECORD_FLOATS), np.float32)
    rec[:, 0:3] = position
    if normal is not None:
        rec[:, 3:6] = normal
    rec[:, 6:54] = np.asarray(sh, np.float32).reshape(n, 48)
    rec[:, 54] = opacity_logit
    rec[:, 55:58] = log_scale
    rec[:, 58:62] = rotation
    props = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)]
             + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {p}\n" for p in props) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode())
        f.write(rec.astype("<f4").tobytes())
