"""Host side of the rasterizer: the role `GSGaussians` plays in the reference
(apps/gsrast/GSGaussians.cpp). PyTorch is used for device memory and streams only; every
stage runs in libgsrast_amd.so through the C ABI (include/gsrast_amd.h).

  ChunkBuffer            <-> resizeFunctional           (GSGaussians.cpp:27-42)
  SplatRasterizer.configure_from_scene <-> configureFromSplatData (:109-153)
  SplatRasterizer.draw   <-> GSGaussians::draw          (:155-212)
  SplatRasterizer.map_geometry_state <-> mapGeometryState (:214-219)
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from .camera import Camera


class ChunkBuffer:
    """Grow-only device chunk with 2x over-allocation, handed to the library as an
    allocator callback: `resizeFunctional` of the reference."""

    def __init__(self, device: torch.device):
        self.device = device
        self.tensor: torch.Tensor | None = None
        self.capacity = 0
        self.calls: list[int] = []          # sizes requested, in order (observable contract)
        self._cb = _capi.ALLOC_FN(self._alloc)

    def _alloc(self, _user, nbytes):
        self.calls.append(int(nbytes))
        try:
            if nbytes > self.capacity:
                self.tensor = None
                self.tensor = torch.empty(2 * int(nbytes), dtype=torch.uint8, device=self.device)
                self.capacity = 2 * int(nbytes)
            return self.tensor.data_ptr()
        except Exception:       # an exception must not cross the C boundary; NULL -> GSR_ERR_ALLOC
            return None

    @property
    def callback(self):
        return self._cb

    def base(self) -> int:
        return 0 if self.tensor is None else self.tensor.data_ptr()

    def view(self, ptr: int, count: int, dtype: torch.dtype) -> torch.Tensor:
        """Typed view of `count` elements starting at device address `ptr` inside the chunk."""
        off = ptr - self.base()
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        assert 0 <= off and off + nbytes <= self.capacity, "state pointer outside its chunk"
        return self.tensor[off:off + nbytes].view(dtype)


def _dev(a, device, dtype=torch.float32) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype).contiguous()


class SplatRasterizer:
    """Uploads a scene once and renders it per camera through gsr_forward."""

    def __init__(self, width: int, height: int, device="cuda:0", background=(0.0, 0.0, 0.0)):
        self.lib = _capi.lib()                       # raises if the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("SplatRasterizer needs a HIP device (no CPU fallback exists)")
        self.device = torch.device(device)
        self.width, self.height = int(width), int(height)
        self.geom = ChunkBuffer(self.device)
        self.binning = ChunkBuffer(self.device)
        self.image = ChunkBuffer(self.device)
        self.background = _dev(np.asarray(background, np.float32), self.device)
        self.out_color = torch.zeros((3, self.height, self.width), dtype=torch.float32, device=self.device)
        self.num_gaussians = 0
        self.use_rects = True
        self.last_num_rendered = 0
        self.last_records_staged = 0
        self.last_plan = "none"
        self.last_blend_from_lists = False
        self.last_lists_written = True
        self.last_receipt: _capi.ForwardReceipt | None = None      # of the last draw(): what backward() / poll take
        self.last_stage_ms: dict[str, float] = {}
        # view (16) | proj (16) | cam_pos (3): one device buffer, uploaded with one async copy from pinned memory
        self._cam_dev = torch.zeros(35, dtype=torch.float32, device=self.device)
        self._cam_host = torch.zeros(35, dtype=torch.float32).pin_memory()
        self._view, self._proj, self._cam_pos = self._cam_dev[0:16], self._cam_dev[16:32], self._cam_dev[32:35]
        self._last_cam = None
        # this view's tile history (gsr_tile_history: how long the tiles of its last frames took; the blend starts the slow
        # ones first). One per rasterizer object, so two of them on one thread do not feed each other's frames.
        self._history = C.c_void_p()
        with torch.cuda.device(self.device):
            _capi.check(self.lib.gsr_tile_history_create(C.byref(self._history)), "gsr_tile_history_create")

    def __del__(self):
        h, self._history = getattr(self, "_history", None), None
        if h is not None and h.value:
            try:
                torch.cuda.synchronize(self.device)          # (the streams it was used on must be idle)
                self.lib.gsr_tile_history_destroy(h)
            except Exception:
                pass

    # -- scene upload -----------------------------------------------------------------
    def configure_from_scene(self, scene: dict, use_rects: bool = True) -> None:
        self.means3D = _dev(scene["means3D"], self.device)
        self.scales = _dev(scene["scales"], self.device)
        self.rotations = _dev(scene["rotations"], self.device)
        self.opacities = _dev(scene["opacities"], self.device)
        self.shs = _dev(scene["shs"], self.device)
        self.num_gaussians = int(self.means3D.shape[0])
        assert self.means3D.shape == (self.num_gaussians, 4) and self.scales.shape == (self.num_gaussians, 4)
        assert self.rotations.shape == (self.num_gaussians, 4) and self.shs.shape == (self.num_gaussians, 48)
        self.use_rects = use_rects
        self._colors_dc, self._colors_key = None, None   # colours precomputed from the DC triples, on first use, per SH tensor state
        self.rects = (torch.zeros((self.num_gaussians, 2), dtype=torch.int32, device=self.device)
                      if use_rects else None)

    def set_camera(self, cam: Camera) -> None:
        """Uploads the 35 camera floats (the reference's caller copies view / proj per frame,
        GSGaussians.cpp:157-176). The same Camera object again is not re-uploaded."""
        assert cam.width == self.width and cam.height == self.height
        if cam is self._last_cam:
            return
        h = self._cam_host.numpy()
        h[0:16] = np.asarray(cam.view, np.float32).reshape(16)
        h[16:32] = np.asarray(cam.proj, np.float32).reshape(16)
        h[32:35] = np.asarray(cam.cam_pos, np.float32).reshape(3)
        self._cam_dev.copy_(self._cam_host, non_blocking=True)     # stream-ordered before the next forward call
        self._tan = (float(cam.tan_fovx), float(cam.tan_fovy))
        self._last_cam = cam

    # -- one frame --------------------------------------------------------------------
    def draw(self, cam: Camera | None = None, *, profile: bool = False, count_staged: bool = False,
             tile_rows: tuple[int, int] | None = None, scale_modifier: float = 1.0,
             sync: bool = True, semantics: str = "gscuda", sh_degree: int = 3, plan: str = "auto",
             overlap_emit: "bool | None" = None, sorted_lists: bool = True, colors_precomp: bool = False,
             tile_history: "bool | str" = True, deep_tiles: "bool | str | None" = None) -> torch.Tensor:
        """One `forward` call on the current torch stream. Returns the planar (3,H,W) image
        tensor owned by this object. `sync` adds the device synchronise the reference's caller
        performs after every call (CudaBuffer.hpp:8-12). semantics="inria" selects the upstream
        rasterizer's semantics (GSR_FLAG_SEMANTICS_INRIA): shs must then be laid out [N][16][3].
        plan: "auto" | "sort" | "blocks" — binning plan (GSR_FLAG_PLAN_*); the one used is in last_plan.
        overlap_emit: True = GSR_FLAG_OVERLAP_EMIT (block plan: the blend on a second stream beside the emission), False =
        GSR_FLAG_SERIAL_EMIT, None = the library decides per call (last_emit_overlapped tells).
        tile_history=False: GSR_FLAG_NO_TILE_HISTORY (the blend does not start the last frame's slowest tiles first); True: this
        object's own gsr_tile_history; "default": none passed — the library's own for the calling thread and stream (what a
        caller of the reference's signature gets); last_tile_order_dropped: the history's frames did not resemble each other.
        sorted_lists=False: GSR_FLAG_NO_SORTED_LISTS (forward-only callers; last_lists_written tells whether the
        binning chunk holds the sorted keys / values of this call).
        deep_tiles: None = the library decides per tile from the history (four waves for the tiles it expects to be the frame's
        slowest, csrc/blend.hip; last_deep_tiles tells whether the blend was launched with them enabled), False =
        GSR_FLAG_NO_DEEP_TILES, "all" = GSR_FLAG_DEEP_TILES_ALL (every tile; a diagnostic), "all8" / "all16" = eight / sixteen
        waves per tile (GSR_FLAG_DEEP_WAVES_8 / _16).
        colors_precomp: pass the scene's colours as the reference's `colorsPrecomp` argument (GSCuda.cuh:111) — computed once
        per scene by gsr_colors_from_dc, bit-equal to what the preprocess writes to geomState.rgb per frame (gscuda semantics
        only: there the colour does not depend on the view)."""
        if cam is not None:
            self.set_camera(cam)
        a = _capi.ForwardArgs()
        a.struct_size = C.sizeof(_capi.ForwardArgs)
        inria = semantics == "inria"
        assert semantics in ("gscuda", "inria")
        a.flags = ((_capi.GSR_FLAG_PROFILE if profile else 0) | (_capi.GSR_FLAG_COUNT_STAGED if count_staged else 0)
                   | (_capi.GSR_FLAG_SEMANTICS_INRIA if inria else 0)
                   | {"auto": 0, "sort": _capi.GSR_FLAG_PLAN_SORT, "blocks": _capi.GSR_FLAG_PLAN_BLOCKS}[plan]
                   | (_capi.GSR_FLAG_OVERLAP_EMIT if overlap_emit else (_capi.GSR_FLAG_SERIAL_EMIT if overlap_emit is False else 0))
                   | (0 if sorted_lists else _capi.GSR_FLAG_NO_SORTED_LISTS)
                   | (0 if tile_history else _capi.GSR_FLAG_NO_TILE_HISTORY)
                   | {"all": _capi.GSR_FLAG_DEEP_TILES_ALL, "all8": _capi.GSR_FLAG_DEEP_WAVES_8, "all16": _capi.GSR_FLAG_DEEP_WAVES_16,
                      False: _capi.GSR_FLAG_NO_DEEP_TILES, None: 0}[deep_tiles])
        a.geometry_alloc, a.binning_alloc, a.image_alloc = self.geom.callback, self.binning.callback, self.image.callback
        a.num_gaussians, a.sh_dims, a.M = self.num_gaussians, (sh_degree if inria else 3), 16
        a.background = self.background.data_ptr()
        a.width, a.height = self.width, self.height
        a.means3D, a.shs = self.means3D.data_ptr(), self.shs.data_ptr()
        a.colors_precomp = self.precomputed_colors().data_ptr() if colors_precomp else None
        assert not (colors_precomp and inria), "upstream colour depends on the view direction"
        a.opacities, a.scales = self.opacities.data_ptr(), self.scales.data_ptr()
        a.scale_modifier = scale_modifier
        a.rotations = self.rotations.data_ptr()
        a.cov3D_precomp = None
        a.view_matrix, a.proj_matrix, a.cam_pos = self._view.data_ptr(), self._proj.data_ptr(), self._cam_pos.data_ptr()
        a.tan_fovx, a.tan_fovy = self._tan
        a.prefiltered = 0
        a.out_color = self.out_color.data_ptr()
        a.radii = None
        a.rects = self.rects.data_ptr() if (self.rects is not None and not inria) else None
        a.box_min = a.box_max = None
        a.stream = torch.cuda.current_stream(self.device).cuda_stream
        a.tile_history = self._history if tile_history is True else None
        if tile_rows is not None:
            a.tile_row_begin, a.tile_row_end = int(tile_rows[0]), int(tile_rows[1])
        with torch.cuda.device(self.device):
            rc = self.lib.gsr_forward(C.byref(a))
        _capi.check(rc, "gsr_forward")
        self.last_receipt = a.receipt.copy()
        self.last_colors_precomp = bool(colors_precomp)
        # which colours THIS call composited (backward() reads the same ones): tied to the call's receipt, not to "the last call"
        self._colors_of_call = (int(a.receipt.serial), self._colors_dc if colors_precomp else None)
        self.last_num_rendered = int(a.num_rendered)
        self.last_records_staged = int(a.records_staged)
        self.last_plan = _capi.PLAN_NAMES[int(a.plan_used) & 0xFF]
        self.last_lists_written = not (int(a.plan_used) & _capi.GSR_PLAN_LISTS_SKIPPED)
        # block plan only: did the blend read the sorted lists (sparse frames) instead of the block lists
        self.last_blend_from_lists = bool(int(a.plan_used) & _capi.GSR_PLAN_BLEND_FROM_LISTS)
        self.last_tiles_reordered = bool(int(a.plan_used) & _capi.GSR_PLAN_TILES_REORDERED)
        self.last_tile_order_dropped = bool(int(a.plan_used) & _capi.GSR_PLAN_TILE_ORDER_DROPPED)
        self.last_emit_overlapped = bool(int(a.plan_used) & _capi.GSR_PLAN_EMIT_OVERLAPPED)
        self.last_colors_beside = bool(int(a.plan_used) & _capi.GSR_PLAN_COLORS_BESIDE)
        self.last_deep_tiles = bool(int(a.plan_used) & _capi.GSR_PLAN_DEEP_TILES)
        self.last_stage_ms = {n: float(a.stage_ms[i]) for i, n in enumerate(_capi.STAGE_NAMES)} if profile else {}
        if sync:
            torch.cuda.current_stream(self.device).synchronize()
            self.poll_async_error()
        return self.out_color

    def tile_history_stats(self) -> dict:
        """gsr_tile_history_stats of this object's history (host side; what the last sort of the blend's tile order found)."""
        out = (C.c_uint32 * 6)()
        _capi.check(self.lib.gsr_tile_history_stats(self._history, out), "gsr_tile_history_stats")
        return {"mean_ticks": int(out[0]), "longest_ticks": int(out[1]), "similarity": int(out[2]) / 1000.0,
                "order_dropped": bool(out[3]), "calls": int(out[4]), "overlapped": bool(out[5])}

    def tile_history_times(self):
        """(times, deep flags, deep count) of this object's history after its last call: numpy u32[tiles] in units of 10 ns,
        bool[tiles] (composited by four waves), and the number of deep tiles of the last sorted order (gsr_tile_history_times)."""
        import numpy as np
        tiles = ((self.width + 15) // 16) * ((self.height + 15) // 16)
        out, deep = (C.c_uint32 * tiles)(), C.c_uint32(0)
        _capi.check(self.lib.gsr_tile_history_times(self._history, out, tiles, C.byref(deep)), "gsr_tile_history_times")
        raw = np.frombuffer(out, dtype=np.uint32).copy()
        return raw & 0x7FFFFFFF, (raw >> 31).astype(bool), int(deep.value)

    def precomputed_colors(self) -> torch.Tensor:
        """vec3[N] = 0.5 + 0.4 DC (gsr_colors_from_dc), computed on first use and kept for the scene."""
        key = (self.shs.data_ptr(), self.shs._version)          # (callers may replace or write rast.shs between frames)
        if self._colors_dc is None or self._colors_key != key:
            c = torch.empty((self.num_gaussians, 3), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                rc = self.lib.gsr_colors_from_dc(self.num_gaussians, self.shs.data_ptr(), c.data_ptr(),
                                                 torch.cuda.current_stream(self.device).cuda_stream)
            _capi.check(rc, "gsr_colors_from_dc")
            self._colors_dc, self._colors_key = c, key
        return self._colors_dc

    def poll_async_error(self, receipt: "_capi.ForwardReceipt | None" = None) -> None:
        """After the stream of a draw() has been synchronised: raises if a device-side wait of that call gave up."""
        r = receipt if receipt is not None else self.last_receipt
        _capi.check(self.lib.gsr_poll_async_error(C.byref(r)), "gsr_forward (device side)")

    # -- point-splat path (gscuda::forwardPoints, GSCuda.cu:110-155) -----------------------------
    def draw_points(self, cam: Camera | None = None, sync: bool = True) -> torch.Tensor:
        """One gsr_forward_points call: every centre lands on one pixel, the nearest wins. The image chunk
        then holds pc::ImageState (depth, temporary image); see map_points_image_state."""
        if cam is not None:
            self.set_camera(cam)
        if getattr(self, "_means3", None) is None or self._means3.shape[0] != self.num_gaussians:
            self._means3 = self.means3D[:, :3].contiguous()        # this path reads a stride of three floats
        a = _capi.ForwardArgs()
        a.struct_size = C.sizeof(_capi.ForwardArgs)
        a.geometry_alloc, a.binning_alloc, a.image_alloc = self.geom.callback, self.binning.callback, self.image.callback
        a.num_gaussians, a.sh_dims, a.M = self.num_gaussians, 3, 16
        a.background = self.background.data_ptr()
        a.width, a.height = self.width, self.height
        a.means3D, a.shs = self._means3.data_ptr(), self.shs.data_ptr()
        a.view_matrix, a.proj_matrix, a.cam_pos = self._view.data_ptr(), self._proj.data_ptr(), self._cam_pos.data_ptr()
        a.tan_fovx, a.tan_fovy = self._tan
        a.out_color = self.out_color.data_ptr()
        a.stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self.lib.gsr_forward_points(C.byref(a))
        _capi.check(rc, "gsr_forward_points")
        if sync:
            torch.cuda.current_stream(self.device).synchronize()
        return self.out_color

    def map_points_image_state(self) -> dict:
        st = _capi.PointsImageState()
        P = self.width * self.height
        self.lib.gsr_points_image_from_chunk(self.image.base(), P, C.byref(st))
        v = self.image.view
        return {"depth": v(st.depth, P, torch.float32).view(self.height, self.width),
                "outColor": v(st.out_color, 3 * P, torch.float32).view(3, self.height, self.width)}

    # -- backward pass (BASELINE config 5; no counterpart in the reference) ------------------
    def backward(self, dL_dout: torch.Tensor, *, profile: bool = False, with_cov3D: bool = True,
                 tile_rows: tuple[int, int] | None = None, scale_modifier: float = 1.0, semantics: str = "gscuda",
                 sh_degree: int = 3, receipt: "_capi.ForwardReceipt | None | bool" = None, wide_sums: bool = True,
                 outputs: "tuple[str, ...] | None" = None) -> dict:
        """Gradients of sum(dL_dout * out_color) of the LAST draw() through gsr_backward; `semantics` / `sh_degree`
        must be those of that draw(). receipt: the gsr_forward_receipt of the draw() this is the backward of (default:
        this object's last draw(); any host thread may call); False = none, the reference's contract only (sorted lists
        in the binning chunk — refused after a draw(sorted_lists=False)). Returns device tensors dL_dmean2D [N,2], dL_dconic_opacity [N,4], dL_dcolors [N,3]
        and, with with_cov3D, dL_dcov2D [N,4] (m00, m01, m11, 0), dL_dcov3D [N,6], dL_dshs [N,48] (gscuda: the DC triple only; inria: every coefficient
        up to sh_degree), dL_dmeans3D / dL_dscales / dL_drotations [N,4].
        wide_sums: accumulate the per-Gaussian sums in double (gsr_backward_args.sums_f64: 96 N bytes of scratch kept by this
        object for as long as it lives — 0.56 GB for the 5.8 M-splat bench scene, 4.8 GB at 50 M; wide_sums=False does without
        it —, zero between calls and dropped if a call fails) — the gradients of screen-filling splats then no longer depend on the order in which the
        tiles' atomics arrive. outputs (needs wide_sums): the names to compute, e.g. BASELINE config 5's ("dL_dmean2D",
        "dL_dcov3D", "dL_dshs"); the others are neither computed nor written (the chain is bound by its writes) and
        absent from the result. The tensors are owned by this object and overwritten by the next call."""
        assert semantics in ("gscuda", "inria")
        n, dev = self.num_gaussians, self.device
        g = dL_dout.to(device=dev, dtype=torch.float32).contiguous()
        assert g.shape == (3, self.height, self.width)
        gst, ist, bst = _capi.GeometryState(), _capi.ImageState(), _capi.BinningState()
        self.lib.gsr_geometry_from_chunk(self.geom.base(), n, C.byref(gst))
        self.lib.gsr_image_from_chunk(self.image.base(), self.width * self.height, C.byref(ist))
        rcpt = self.last_receipt if receipt is None else (None if receipt is False else receipt)
        num_rendered = int(rcpt.num_rendered) if rcpt is not None else self.last_num_rendered
        self.lib.gsr_binning_from_chunk(self.binning.base(), num_rendered, C.byref(bst))
        # output buffers are allocated once per scene and semantics and reused. dL_dshs is 48 floats per Gaussian; the
        # gscuda chain writes floats 0..15 of every Gaussian, the inria chain all 48: the buffer is zero-filled once, and
        # kept per semantics so that a gscuda call never returns what an inria call left in floats 16..47.
        caches = self.__dict__.setdefault("_bw_out_by_semantics", {})
        cache = caches.get(semantics)
        if cache is None or cache["dL_dmean2D"].shape[0] != n or ("dL_dcov3D" in cache) != with_cov3D:
            cache = {"dL_dmean2D": torch.empty((n, 2), dtype=torch.float32, device=dev),
                     "dL_dconic_opacity": torch.empty((n, 4), dtype=torch.float32, device=dev),
                     "dL_dcolors": torch.empty((n, 3), dtype=torch.float32, device=dev)}
            if with_cov3D:
                cache["dL_dcov2D"] = torch.empty((n, 4), dtype=torch.float32, device=dev)     # (m00, m01, m11, 0): what the chain starts from
                cache["dL_dcov3D"] = torch.empty((n, 6), dtype=torch.float32, device=dev)
                cache["dL_dshs"] = torch.zeros((n, 48), dtype=torch.float32, device=dev)
                for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations"):
                    cache[k] = torch.empty((n, 4), dtype=torch.float32, device=dev)
            caches[semantics] = cache
        out = cache
        a = _capi.BackwardArgs()
        a.struct_size = C.sizeof(_capi.BackwardArgs)
        a.flags = (_capi.GSR_FLAG_PROFILE if profile else 0) | (_capi.GSR_FLAG_SEMANTICS_INRIA if semantics == "inria" else 0)
        if semantics == "inria":
            a.cam_pos, a.shs, a.clamped, a.sh_dims = self._cam_pos.data_ptr(), self.shs.data_ptr(), gst.clamped, int(sh_degree)
        a.num_gaussians, a.width, a.height = n, self.width, self.height
        a.background = self.background.data_ptr()
        a.means2D, a.conic_opacity, a.cov3D = gst.means2D, gst.conic_opacity, gst.cov3D
        serial, col = getattr(self, "_colors_of_call", (None, None))
        if rcpt is not None and int(rcpt.serial) != serial:
            col = None                               # a receipt of another call: that call's colours are its geomState.rgb
        a.colors = col.data_ptr() if col is not None else gst.rgb
        a.radii = gst.internal_radii
        a.ranges, a.n_contrib, a.final_t = ist.ranges, ist.n_contrib, ist.accum_alpha
        a.point_list = bst.values
        a.means3D, a.view_matrix = self.means3D.data_ptr(), self._view.data_ptr()
        a.tan_fovx, a.tan_fovy = self._tan
        a.dL_dout_color = g.data_ptr()
        if outputs is not None:
            assert wide_sums and set(outputs) <= set(cache), (outputs, sorted(cache))
            out = {k: cache[k] for k in outputs}
        ptr = lambda k: out[k].data_ptr() if k in out else None
        a.dL_dmean2D, a.dL_dconic_opacity = ptr("dL_dmean2D"), ptr("dL_dconic_opacity")
        a.dL_dcolors = ptr("dL_dcolors")
        a.dL_dcov3D = ptr("dL_dcov3D")
        a.dL_dcov2D = ptr("dL_dcov2D")
        if wide_sums:
            if getattr(self, "_sums_f64", None) is None or self._sums_f64.shape[0] != n:
                self._sums_f64 = torch.zeros((n, 12), dtype=torch.float64, device=dev)      # (the library leaves it zero)
            a.sums_f64 = self._sums_f64.data_ptr()
        a.dL_dshs = ptr("dL_dshs")
        if with_cov3D:
            a.proj_matrix, a.scales, a.rotations = self._proj.data_ptr(), self.scales.data_ptr(), self.rotations.data_ptr()
            a.scale_modifier = scale_modifier
            a.dL_dmeans3D, a.dL_dscales = ptr("dL_dmeans3D"), ptr("dL_dscales")
            a.dL_drotations = ptr("dL_drotations")
        a.stream = torch.cuda.current_stream(dev).cuda_stream
        if tile_rows is not None:
            a.tile_row_begin, a.tile_row_end = int(tile_rows[0]), int(tile_rows[1])
        if rcpt is not None:
            a.receipt = rcpt
        with torch.cuda.device(dev):
            rc = self.lib.gsr_backward(C.byref(a))
        if rc != _capi.GSR_OK and wide_sums:
            self._sums_f64 = None           # (a call that failed half way may have left sums behind: the next call starts from a zeroed scratch again)
        _capi.check(rc, "gsr_backward")
        self.last_backward_ms = (float(a.stage_ms[0]), float(a.stage_ms[1])) if profile else ()
        torch.cuda.current_stream(dev).synchronize()
        return out

    # -- state inspection (what the reference's Inspector does through fromChunk) ------
    def map_geometry_state(self) -> dict:
        st = _capi.GeometryState()
        self.lib.gsr_geometry_from_chunk(self.geom.base(), self.num_gaussians, C.byref(st))
        n, v = self.num_gaussians, self.geom.view
        return {
            "tilesTouched": v(st.tiles_touched, n, torch.int32),
            "depths": v(st.depths, n, torch.float32),
            "radii": v(st.internal_radii, n, torch.int32),
            "means2D": v(st.means2D, 2 * n, torch.float32).view(n, 2),
            "cov3D": v(st.cov3D, 6 * n, torch.float32).view(n, 6),
            "conicOpacity": v(st.conic_opacity, 4 * n, torch.float32).view(n, 4),
            "rgb": v(st.rgb, 3 * n, torch.float32).view(n, 3),
            "pointOffsets": v(st.point_offsets, n, torch.int32),
        }

    def map_image_state(self) -> dict:
        st = _capi.ImageState()
        P = self.width * self.height
        self.lib.gsr_image_from_chunk(self.image.base(), P, C.byref(st))
        T = ((self.width + 15) // 16) * ((self.height + 15) // 16)
        v = self.image.view
        return {
            "ranges": v(st.ranges, 2 * T, torch.int32).view(T, 2),
            "nContrib": v(st.n_contrib, P, torch.int32).view(self.height, self.width),
            "finalT": v(st.accum_alpha, P, torch.float32).view(self.height, self.width),
        }

    def map_binning_state(self) -> dict:
        st = _capi.BinningState()
        R = self.last_num_rendered
        self.lib.gsr_binning_from_chunk(self.binning.base(), R, C.byref(st))
        v = self.binning.view
        return {
            "keys_unsorted": v(st.keys_unsorted, R, torch.int64),
            "keys": v(st.keys, R, torch.int64),
            "values_unsorted": v(st.values_unsorted, R, torch.int32),
            "values": v(st.values, R, torch.int32),
        }


# ---- stage-level wrappers (unit parity tests) ------------------------------------------
def inclusive_scan_u32(x: torch.Tensor) -> torch.Tensor:
    L = _capi.lib()
    assert x.dtype == torch.int32 and x.is_cuda and x.is_contiguous()
    out = torch.empty_like(x)
    temp = torch.empty(max(int(L.gsr_scan_temp_bytes(x.numel())), 128), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.gsr_inclusive_scan_u32(x.data_ptr(), out.data_ptr(), x.numel(), temp.data_ptr(),
                                      torch.cuda.current_stream(x.device).cuda_stream)
    _capi.check(rc, "gsr_inclusive_scan_u32")
    torch.cuda.current_stream(x.device).synchronize()
    return out


def sort_pairs(keys: torch.Tensor, values: torch.Tensor, end_bit: int = 64, begin_bit: int = 0, sync: bool = True,
               out=None, temp=None):
    L = _capi.lib()
    assert keys.dtype == torch.int64 and values.dtype == torch.int32 and keys.is_cuda
    assert keys.is_contiguous() and values.is_contiguous() and keys.numel() == values.numel()
    n = keys.numel()
    ko, vo = out if out is not None else (torch.empty_like(keys), torch.empty_like(values))
    if temp is None:
        temp = torch.empty(max(int(L.gsr_sort_temp_bytes(n)), 128), dtype=torch.uint8, device=keys.device)
    with torch.cuda.device(keys.device):
        rc = L.gsr_sort_pairs_u64_u32(keys.data_ptr(), ko.data_ptr(), values.data_ptr(), vo.data_ptr(), n, begin_bit,
                                      end_bit, temp.data_ptr(), torch.cuda.current_stream(keys.device).cuda_stream)
    _capi.check(rc, "gsr_sort_pairs_u64_u32")
    if sync:
        torch.cuda.current_stream(keys.device).synchronize()
    return ko, vo
