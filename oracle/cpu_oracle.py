"""ctypes front-end of the C++ oracle (oracle/gsr_oracle.cpp).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. Parity is unpinned by the reference (see the .cpp header).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgsr_oracle.so")
# the same source built with -ffp-contract=fast -mfma: contraction sensitivity study only (oracle/Makefile)
_LIB_PATH_CONTRACT = os.path.join(_HERE, "_build", "libgsr_oracle_contract.so")
_libs = {}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gsr_oracle.cpp")
    stale = any(not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src) for p in (_LIB_PATH, _LIB_PATH_CONTRACT))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def lib(contract: bool = False) -> ctypes.CDLL:
    if contract not in _libs:
        build()
        L = ctypes.CDLL(_LIB_PATH_CONTRACT if contract else _LIB_PATH)
        L.gsro_preprocess.restype = ctypes.c_uint64
        L.gsro_preprocess_mt.restype = ctypes.c_uint64
        L.gsro_blend.restype = ctypes.c_uint64
        L.gsro_higher_msb.restype = ctypes.c_uint32
        L.gsro_hardware_concurrency.restype = ctypes.c_uint
        _libs[contract] = L
    return _libs[contract]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def higher_msb(n: int) -> int:
    return int(lib().gsro_higher_msb(ctypes.c_uint32(n)))


def expf(x: np.ndarray) -> np.ndarray:
    """exp of float32 arguments as the oracle's tile loop computes it (libm's expf)."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().gsro_expf(ctypes.c_long(x.size), _p(x), _p(out))
    return out


def hardware_concurrency() -> int:
    return int(lib().gsro_hardware_concurrency())


def forward(scene: dict, cam, background=(0.0, 0.0, 0.0), use_rects: bool = True, scale_modifier: float = 1.0,
            threads: int = 1, out_init: np.ndarray | None = None, timings: dict | None = None,
            contract: bool = False) -> dict:
    """Runs the whole reference pipeline (GSCuda.cu:695-811) on the CPU and returns every
    intermediate: GeometryState arrays, rects, keys/values (unsorted and sorted), tile
    ranges, the planar image, finalT, nContrib, R (num_rendered) and R_f (records staged).
    Arrays the reference leaves unwritten keep their zero initialisation.
    contract=True runs the build with fused multiply-adds allowed (contraction sensitivity study only)."""
    L = lib(contract)
    n = int(scene["means3D"].shape[0])
    W, H = cam.width, cam.height
    P = W * H
    gx, gy = (W + 15) // 16, (H + 15) // 16
    f32 = lambda k: np.ascontiguousarray(scene[k], dtype=np.float32)
    means, scales, rots, opac, shs = f32("means3D"), f32("scales"), f32("rotations"), f32("opacities"), f32("shs")
    view = np.ascontiguousarray(cam.view, dtype=np.float32)
    proj = np.ascontiguousarray(cam.proj, dtype=np.float32)
    o = {
        "radii": np.zeros(n, np.int32), "means2D": np.zeros((n, 2), np.float32),
        "depths": np.zeros(n, np.float32), "cov3D": np.zeros((n, 6), np.float32),
        "rgb": np.zeros((n, 3), np.float32), "conicOpacity": np.zeros((n, 4), np.float32),
        "tilesTouched": np.zeros(n, np.uint32), "pointOffsets": np.zeros(n, np.uint32),
        "rects": np.zeros((n, 2), np.int32) if use_rects else None,
    }
    t0 = time.perf_counter()
    R = int(L.gsro_preprocess_mt(
        ctypes.c_int(n), _p(means), _p(scales), ctypes.c_float(scale_modifier), _p(rots), _p(opac), _p(shs),
        None, None, _p(view), _p(proj), ctypes.c_int(W), ctypes.c_int(H),
        ctypes.c_float(cam.tan_fovx), ctypes.c_float(cam.tan_fovy),
        _p(o["radii"]), _p(o["means2D"]), _p(o["depths"]), _p(o["cov3D"]), _p(o["rgb"]),
        _p(o["conicOpacity"]), _p(o["tilesTouched"]), _p(o["rects"]), _p(o["pointOffsets"]), ctypes.c_int(threads)))
    t1 = time.perf_counter()
    o["num_rendered"] = R
    bg = np.asarray(background, dtype=np.float32)
    o["out_color"] = (np.zeros((3, H, W), np.float32) if out_init is None
                      else np.ascontiguousarray(out_init, dtype=np.float32).copy())
    o["finalT"] = np.zeros((H, W), np.float32)
    o["nContrib"] = np.zeros((H, W), np.uint32)
    o["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    o["keys_unsorted"] = np.zeros(R, np.uint64)
    o["values_unsorted"] = np.zeros(R, np.uint32)
    o["keys"] = np.zeros(R, np.uint64)
    o["values"] = np.zeros(R, np.uint32)
    o["records_staged"] = 0
    t2 = t3 = t1
    if R > 0:   # GSCuda.cu:775-778: R == 0 returns before anything else is written
        # (the stable sort runs on `threads` host threads: same result as the single-thread std::stable_sort)
        L.gsro_bin_mt(ctypes.c_int(n), ctypes.c_int(W), ctypes.c_int(H), _p(o["radii"]), _p(o["means2D"]),
                      _p(o["depths"]), _p(o["pointOffsets"]), _p(o["rects"]), ctypes.c_uint64(R),
                      _p(o["keys_unsorted"]), _p(o["values_unsorted"]), _p(o["keys"]), _p(o["values"]),
                      _p(o["ranges"]), ctypes.c_int(threads))
        t2 = time.perf_counter()
        o["records_staged"] = int(L.gsro_blend(
            ctypes.c_int(W), ctypes.c_int(H), _p(o["ranges"]), _p(o["values"]), _p(o["means2D"]), _p(o["rgb"]),
            _p(o["conicOpacity"]), _p(bg), _p(o["finalT"]), _p(o["nContrib"]), _p(o["out_color"]),
            ctypes.c_int(threads)))
        t3 = time.perf_counter()
    if timings is not None:
        timings.update(preprocess_s=t1 - t0, bin_s=t2 - t1, blend_s=t3 - t2, total_s=t3 - t0)
    return o


def blend_only(state: dict, cam, background=(0.0, 0.0, 0.0), threads: int = 1):
    """Re-runs only the tile loop (GSCuda.cu:543-677) on an existing oracle state.
    Returns (seconds, records_staged); used as the timed CPU baseline."""
    L = lib()
    W, H = cam.width, cam.height
    bg = np.asarray(background, dtype=np.float32)
    out = np.zeros((3, H, W), np.float32)
    ft = np.zeros((H, W), np.float32)
    nc = np.zeros((H, W), np.uint32)
    t0 = time.perf_counter()
    staged = int(L.gsro_blend(ctypes.c_int(W), ctypes.c_int(H), _p(state["ranges"]), _p(state["values"]),
                              _p(state["means2D"]), _p(state["rgb"]), _p(state["conicOpacity"]), _p(bg),
                              _p(ft), _p(nc), _p(out), ctypes.c_int(threads)))
    return time.perf_counter() - t0, staged
