"""ctypes binding of include/gsrast_amd.h (the C ABI of libgsrast_amd.so).

There is no fallback: if the HIP library has not been built, importing the product path
raises. Field order and types below mirror the header exactly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSR_LIB_TAG selects a tuning build made with `python -m gsrast_amd.build --tag NAME` (experiments only).
_TAG = os.environ.get("GSR_LIB_TAG", "")
LIB_PATH = os.path.join(_HERE, "lib", f"libgsrast_amd{'_' + _TAG if _TAG else ''}.so")

GSR_OK = 0
GSR_ERR_INVALID_ARG = 1
GSR_ERR_ALLOC = 2
GSR_ERR_HIP = 3
GSR_ERR_NO_DEVICE = 4
GSR_ERR_TOO_LARGE = 5
GSR_ERR_INTERNAL = 6
GSR_ERR_STALE_RECEIPT = 7

GSR_FLAG_PROFILE = 0x1
GSR_FLAG_COUNT_STAGED = 0x2
GSR_FLAG_SEMANTICS_INRIA = 0x4
GSR_FLAG_PLAN_SORT = 0x8
GSR_FLAG_PLAN_BLOCKS = 0x10
GSR_FLAG_OVERLAP_EMIT = 0x20
GSR_FLAG_NO_SORTED_LISTS = 0x40
GSR_FLAG_NO_TILE_HISTORY = 0x80
GSR_FLAG_SERIAL_EMIT = 0x100
GSR_FLAG_NO_DEEP_TILES = 0x200
GSR_FLAG_DEEP_TILES_ALL = 0x400
GSR_FLAG_DEEP_WAVES_8 = 0x800
GSR_FLAG_DEEP_WAVES_16 = 0x1000
GSR_PLAN_LISTS_SKIPPED = 0x100
GSR_PLAN_BLEND_FROM_LISTS = 0x200
GSR_PLAN_TILES_REORDERED = 0x400
GSR_PLAN_EMIT_OVERLAPPED = 0x800
GSR_PLAN_COLORS_BESIDE = 0x1000
GSR_PLAN_TILE_ORDER_DROPPED = 0x2000
GSR_PLAN_DEEP_TILES = 0x4000
GSR_SH_LAYOUT_FILE, GSR_SH_LAYOUT_COEFFICIENT_MAJOR = 0, 1
PLAN_NAMES = {0: "none", 1: "sort", 2: "blocks", 3: "generic"}
GSR_NUM_STAGES = 8
STAGE_NAMES = ("preprocess", "scan", "depth_order", "duplicate", "sort_pass1", "sort_pass2", "ranges", "blend")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class GeometryState(C.Structure):
    _fields_ = [("tiles_touched", C.c_void_p), ("scan_size", C.c_size_t), ("num_rendered", C.c_uint32),
                ("scanning_space", C.c_void_p), ("depths", C.c_void_p), ("clamped", C.c_void_p),
                ("internal_radii", C.c_void_p), ("means2D", C.c_void_p), ("cov3D", C.c_void_p),
                ("conic_opacity", C.c_void_p), ("rgb", C.c_void_p), ("point_offsets", C.c_void_p)]


class ImageState(C.Structure):
    _fields_ = [("ranges", C.c_void_p), ("n_contrib", C.c_void_p), ("accum_alpha", C.c_void_p)]


class BinningState(C.Structure):
    _fields_ = [("keys_unsorted", C.c_void_p), ("keys", C.c_void_p), ("values_unsorted", C.c_void_p),
                ("values", C.c_void_p), ("sorting_size", C.c_size_t), ("sorting_space", C.c_void_p)]


GSR_RECEIPT_MAGIC = 0x31525347
GSR_LISTS_SKIPPED_STAMP = 0xFFFFFFFF


class ForwardReceipt(C.Structure):
    """gsr_forward_receipt: plain data, copied by value (ctypes copies nested structures on assignment)."""
    _fields_ = [("magic", C.c_uint32), ("plan_used", C.c_uint32),
                ("num_gaussians", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("tile_row_begin", C.c_int32), ("tile_row_end", C.c_int32),
                ("num_rendered", C.c_uint32), ("num_visible", C.c_uint32), ("serial", C.c_uint32),
                ("geometry_chunk", C.c_void_p), ("image_chunk", C.c_void_p), ("binning_chunk", C.c_void_p),
                ("async_words", C.c_void_p), ("tile_history", C.c_void_p)]

    def copy(self) -> "ForwardReceipt":
        r = ForwardReceipt()
        C.memmove(C.byref(r), C.byref(self), C.sizeof(ForwardReceipt))
        return r


class ForwardArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("flags", C.c_uint32),
        ("geometry_alloc", ALLOC_FN), ("geometry_user", C.c_void_p),
        ("binning_alloc", ALLOC_FN), ("binning_user", C.c_void_p),
        ("image_alloc", ALLOC_FN), ("image_user", C.c_void_p),
        ("num_gaussians", C.c_int32), ("sh_dims", C.c_int32), ("M", C.c_int32),
        ("background", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
        ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("scale_modifier", C.c_float),
        ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("view_matrix", C.c_void_p),
        ("proj_matrix", C.c_void_p), ("cam_pos", C.c_void_p), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
        ("prefiltered", C.c_int32), ("out_color", C.c_void_p), ("radii", C.c_void_p), ("rects", C.c_void_p),
        ("box_min", C.c_void_p), ("box_max", C.c_void_p),
        ("stream", C.c_void_p), ("tile_row_begin", C.c_int32), ("tile_row_end", C.c_int32),
        ("tile_history", C.c_void_p),
        ("num_rendered", C.c_uint32), ("records_staged", C.c_uint64),
        ("stage_ms", C.c_float * GSR_NUM_STAGES),
        ("plan_used", C.c_uint32),
        ("receipt", ForwardReceipt),
    ]


class PointsImageState(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("out_color", C.c_void_p), ("default_depth", C.c_void_p), ("winner", C.c_void_p)]


class BackwardArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("flags", C.c_uint32),
        ("num_gaussians", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("background", C.c_void_p),
        ("means2D", C.c_void_p), ("conic_opacity", C.c_void_p), ("colors", C.c_void_p), ("cov3D", C.c_void_p),
        ("radii", C.c_void_p), ("ranges", C.c_void_p), ("n_contrib", C.c_void_p), ("final_t", C.c_void_p),
        ("point_list", C.c_void_p),
        ("means3D", C.c_void_p), ("view_matrix", C.c_void_p), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
        ("dL_dout_color", C.c_void_p),
        ("dL_dmean2D", C.c_void_p), ("dL_dconic_opacity", C.c_void_p), ("dL_dcolors", C.c_void_p),
        ("dL_dcov3D", C.c_void_p), ("dL_dshs", C.c_void_p), ("dL_dcov2D", C.c_void_p),
        ("sums_f64", C.c_void_p),
        ("proj_matrix", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("scale_modifier", C.c_float),
        ("dL_dmeans3D", C.c_void_p), ("dL_dscales", C.c_void_p), ("dL_drotations", C.c_void_p),
        ("stream", C.c_void_p), ("tile_row_begin", C.c_int32), ("tile_row_end", C.c_int32),
        ("stage_ms", C.c_float * 2),
        ("cam_pos", C.c_void_p), ("shs", C.c_void_p), ("clamped", C.c_void_p), ("sh_dims", C.c_int32),
        ("receipt", ForwardReceipt),
    ]


# name -> (restype, argtypes); this is also the list the symbol test checks against the header.
SIGNATURES = {
    "gsr_geometry_from_chunk": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(GeometryState)]),
    "gsr_image_from_chunk": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(ImageState)]),
    "gsr_binning_from_chunk": (C.c_void_p, [C.c_void_p, C.c_size_t, C.POINTER(BinningState)]),
    "gsr_required_geometry": (C.c_size_t, [C.c_int]),
    "gsr_required_image": (C.c_size_t, [C.c_int]),
    "gsr_required_binning": (C.c_size_t, [C.c_size_t]),
    "gsr_forward": (C.c_int, [C.POINTER(ForwardArgs)]),
    "gsr_backward": (C.c_int, [C.POINTER(BackwardArgs)]),
    "gsr_points_image_from_chunk": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(PointsImageState)]),
    "gsr_required_points_image": (C.c_size_t, [C.c_int]),
    "gsr_forward_points": (C.c_int, [C.POINTER(ForwardArgs)]),
    "gsr_last_error": (C.c_int, []),
    "gsr_error_string": (C.c_char_p, [C.c_int]),
    "gsr_last_hip_error": (C.c_char_p, []),
    "gsr_poll_async_error": (C.c_int, [C.POINTER(ForwardReceipt)]),
    "gsr_tile_history_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gsr_tile_history_destroy": (C.c_int, [C.c_void_p]),
    "gsr_tile_history_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "gsr_tile_history_forget_stream": (C.c_int, [C.c_void_p]),
    "gsr_tile_history_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]),
    "gsr_thread_release": (C.c_int, []),
    "gsr_reread_environment": (None, []),
    "gsr_device_shape": (None, [C.c_int, C.POINTER(C.c_uint32)]),
    "gsr_higher_msb": (C.c_uint32, [C.c_uint32]),
    "gsr_scan_temp_bytes": (C.c_size_t, [C.c_size_t]),
    "gsr_inclusive_scan_u32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "gsr_sort_temp_bytes": (C.c_size_t, [C.c_size_t]),
    "gsr_sort_pairs_u64_u32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]),
    "gsr_exchange_unique_id": (C.c_int, [C.c_char_p, C.c_char_p]),
    "gsr_exchange_create": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gsr_exchange_bands": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "gsr_exchange_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int,
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gsr_exchange_loopback": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "gsr_exchange_destroy": (C.c_int, [C.c_void_p]),
    "gsr_exchange_last_error": (C.c_char_p, []),
    "gsr_ply_parse_header": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "gsr_ply_activate": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "gsr_colors_from_dc": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_blend_expf": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_footprint_misses_tile": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gsr_ply_activate_layout": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p]),
}

_lib = None


class GsrError(RuntimeError):
    def __init__(self, code: int, where: str):
        L = lib()
        msg = L.gsr_error_string(code).decode()
        hip = L.gsr_last_hip_error().decode()
        super().__init__(f"{where}: {msg} (code {code})" + (f" [{hip}]" if hip else ""))
        self.code = code


def lib() -> C.CDLL:
    """Loads libgsrast_amd.so. Raises if it has not been built: there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP library first (python -m gsrast_amd.build). "
                "gsrast_amd has no CPU or PyTorch fallback.")
        # PyTorch bundles its own libamdhip64.so.7; it must be the one already mapped when our
        # library resolves that soname, otherwise two HIP runtimes end up in the process and
        # torch's device pointers mean nothing to ours ("no HIP device").
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is None and _TAG:
                continue                       # (a tagged A/B build of an older tree may lack the newest entry points)
            if fn is None:
                raise RuntimeError(f"{LIB_PATH} does not export {name}: rebuild it (python -m gsrast_amd.build)")
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code: int, where: str) -> None:
    if code != GSR_OK:
        raise GsrError(code, where)
