"""Multi-GPU sharding of one frame by screen tile rows (nothing of this exists
in the single-GPU reference). One process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

  * Scene: rank 0's SoA is broadcast once (broadcast_scene); every rank keeps all N Gaussians
    and runs preprocess on all of them (cheaper than exchanging 2-D records per frame).
  * Frame: rank g bins / sorts / blends only tile rows [rows[g], rows[g+1]) — the library clips
    every Gaussian's rectangle to the band (gsr_forward_args.tile_row_begin/end), so per-tile
    lists, and therefore pixels, are identical to the single-GPU frame.
  * Exchange: every rank sends its planar row band to every peer and receives theirs straight into
    its frame (one group of exact-size point-to-point transfers per frame; no staging, no padding);
    no other data-path communication. On RCCL the group is issued by ONE C call into the library
    (gsr_exchange_bands: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a communicator of its own) —
    the same exchange through torch.distributed costs hundreds of microseconds of host time per frame;
    that path remains as the fallback (and is what the gloo tests run). root=r: only rank r receives.
  * Balance: bands are re-cut between frames from the per-tile-row instance counts of the last
    frame (each rank knows its own rows; one small all-gather of grid_y integers).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

TILE = 16


def uniform_bands(grid_y: int, world: int) -> list[int]:
    """world+1 boundaries splitting grid_y tile rows as evenly as possible."""
    return [(grid_y * g) // world for g in range(world + 1)]


def balanced_bands(row_cost: np.ndarray, world: int, floor_cost: float = 0.0) -> list[int]:
    """Boundaries that equalise the summed cost of contiguous tile rows. `floor_cost` is a
    per-row constant (pixel work that exists even for empty rows). Every band keeps >= 1 row
    when grid_y >= world."""
    cost = np.asarray(row_cost, dtype=np.float64) + float(floor_cost)
    grid_y = cost.size
    if grid_y <= world:
        return [min(g, grid_y) for g in range(world + 1)]
    prefix = np.concatenate([[0.0], np.cumsum(cost)])
    total = prefix[-1]
    if total <= 0:
        return uniform_bands(grid_y, world)
    bounds = [0]
    for g in range(1, world):
        target = total * g / world
        b = int(np.searchsorted(prefix, target, side="left"))
        # choose the closer of b-1 / b, keep strictly increasing and leave room for the rest
        if b > 0 and abs(prefix[b - 1] - target) <= abs(prefix[min(b, grid_y)] - target):
            b -= 1
        b = max(b, bounds[-1] + 1)
        b = min(b, grid_y - (world - g))
        bounds.append(b)
    bounds.append(grid_y)
    return bounds


def band_pixel_rows(bounds: list[int], g: int, height: int) -> tuple[int, int]:
    return min(bounds[g] * TILE, height), min(bounds[g + 1] * TILE, height)


def broadcast_scene(scene: dict | None, device, src: int = 0) -> dict:
    """Rank `src` passes its scene (dict of arrays); every rank returns device tensors."""
    rank = dist.get_rank()
    keys = ["means3D", "scales", "rotations", "opacities", "shs"]
    if rank == src:
        n = torch.tensor([int(scene["means3D"].shape[0])], dtype=torch.int64, device=device)
    else:
        n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    n = int(n.item())
    shapes = {"means3D": (n, 4), "scales": (n, 4), "rotations": (n, 4), "opacities": (n,), "shs": (n, 48)}
    out = {}
    for k in keys:
        if rank == src:
            v = scene[k]
            t = (v.to(device=device, dtype=torch.float32).contiguous() if isinstance(v, torch.Tensor)
                 else torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).to(device))
        else:
            t = torch.empty(shapes[k], dtype=torch.float32, device=device)
        dist.broadcast(t, src)
        out[k] = t
    return out


def _torch_rccl_path() -> str | None:
    """The librccl PyTorch has mapped (the library's own dlopen of it then yields the same instance)."""
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


class RowBandExchange:
    """Exchanges planar (3,H,W) row bands so that every rank (or only `root`) ends up with the full frame.

    A band of a planar CHW frame is three contiguous pieces (rows [y0, y1) of each colour plane), so the
    exchange is one group of point-to-point transfers of exact sizes — this rank's three pieces to every
    peer, every peer's three pieces received straight into their place in the frame: no staging buffer,
    no padding to the tallest band, no copy back. Each of the seven xGMI links of a fully connected node
    carries one peer's band.

    transport: "rccl" — the library's own communicator, one C call per frame (gsr_exchange_bands); "torch" —
    torch.distributed.batch_isend_irecv with the op list kept between frames; "auto" (default; GSR_EXCHANGE overrides) —
    rccl when the process group runs on nccl and the library's exchange passes a self-check against the expected frame
    on every rank, else torch. root: None = every rank receives every band; r = only rank r does."""

    def __init__(self, width: int, height: int, device, root: int | None = None, transport: str = "auto"):
        self.width, self.height = width, height
        self.grid_y = (height + TILE - 1) // TILE
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.root = root
        self.bounds = uniform_bands(self.grid_y, self.world)
        self._ops_key, self._ops = None, []
        self._handle = None                          # gsr_exchange*
        self._bounds_c = None
        self.transport = "torch"
        self.transport_note = ""
        want = os.environ.get("GSR_EXCHANGE", transport)
        assert want in ("auto", "torch", "rccl"), want
        # (composite backend strings — "cuda:nccl,cpu:gloo" — name nccl too)
        if want != "torch" and self.world > 1 and self.device.type == "cuda" and "nccl" in str(dist.get_backend()):
            self._try_rccl(must=want == "rccl")

    # -- the library's own communicator -------------------------------------------------------------------------------
    def _try_rccl(self, must: bool) -> None:
        from . import _capi
        L = _capi.lib()
        path = _torch_rccl_path()
        cpath = path.encode() if path else None
        ok = torch.ones(1, dtype=torch.int32, device=self.device)
        ident = torch.zeros(128, dtype=torch.uint8, device=self.device)
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            if L.gsr_exchange_unique_id(cpath, buf) == _capi.GSR_OK:
                ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(self.device)
            else:
                ok.zero_()
        dist.broadcast(ident, 0)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        note = ""
        if int(ok.item()) == 1:
            handle = C.c_void_p()
            with torch.cuda.device(self.device):
                rc = L.gsr_exchange_create(cpath, bytes(ident.cpu().numpy().tobytes()), self.rank, self.world, C.byref(handle))
            if rc != _capi.GSR_OK:
                ok.zero_()
                note = L.gsr_exchange_last_error().decode()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                self._handle = handle
                self.transport = "rccl"
                # self-check: every rank fills its band of a frame whose every row is known, exchanges, and compares
                good = self._self_check()
                flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) != 1:
                    note = "self-check of gsr_exchange_bands failed on some rank"
                    L.gsr_exchange_destroy(self._handle)
                    self._handle, self.transport = None, "torch"
            elif handle.value:
                L.gsr_exchange_destroy(handle)
        else:
            note = "ncclGetUniqueId failed on rank 0: " + L.gsr_exchange_last_error().decode()
        self.transport_note = note
        if must and self.transport != "rccl":
            raise RuntimeError("GSR_EXCHANGE=rccl: " + (note or "the library's RCCL exchange is not available"))

    def _expected_frame(self) -> torch.Tensor:
        ys = torch.arange(self.height, dtype=torch.float32, device=self.device)[None, :, None]
        return (torch.arange(3, dtype=torch.float32, device=self.device)[:, None, None] * 4096.0 + ys).expand(3, self.height, self.width).contiguous()

    def _self_check(self) -> bool:
        want = self._expected_frame()
        good = True
        for bounds in (uniform_bands(self.grid_y, self.world), [0] + [self.grid_y] * self.world):      # (the second: all but rank 0 empty)
            saved = self.bounds
            self.set_bounds(bounds)
            y0, y1 = band_pixel_rows(self.bounds, self.rank, self.height)
            frame = torch.full((3, self.height, self.width), -1.0, device=self.device)
            frame[:, y0:y1, :] = want[:, y0:y1, :]
            self.gather(frame)
            torch.cuda.current_stream(self.device).synchronize()
            if self.root is None or self.rank == self.root:
                good = good and bool(torch.equal(frame, want))
            self.set_bounds(saved)
        return good

    def close(self) -> None:
        if self._handle is not None:
            from . import _capi
            _capi.lib().gsr_exchange_destroy(self._handle)
            self._handle = None

    # -- bands ------------------------------------------------------------------------------------------------------
    def set_bounds(self, bounds: list[int]) -> None:
        assert len(bounds) == self.world + 1 and bounds[0] == 0 and bounds[-1] == self.grid_y
        assert all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
        self.bounds = list(bounds)
        self._bounds_c = (C.c_int32 * (self.world + 1))(*self.bounds)
        self._ops_key = None

    def my_tile_rows(self) -> tuple[int, int]:
        return self.bounds[self.rank], self.bounds[self.rank + 1]

    def _torch_ops(self, frame: torch.Tensor) -> list:
        """The group's P2POps for this frame buffer and these bands: built when either changes, reused every frame."""
        key = (frame.data_ptr(), tuple(self.bounds))
        if key != self._ops_key:
            y0, y1 = band_pixel_rows(self.bounds, self.rank, self.height)
            ops = []
            for step in range(1, self.world):
                # staggered peers: in round `step` rank r sends to r + step and receives from r - step
                dst, src = (self.rank + step) % self.world, (self.rank - step) % self.world
                a, b = band_pixel_rows(self.bounds, src, self.height)
                i_send = y1 > y0 and (self.root is None or dst == self.root)
                i_recv = b > a and (self.root is None or self.rank == self.root)
                for c in range(3):
                    if i_send:
                        ops.append(dist.P2POp(dist.isend, frame[c, y0:y1, :], dst))
                    if i_recv:
                        ops.append(dist.P2POp(dist.irecv, frame[c, a:b, :], src))
            self._ops_key, self._ops = key, ops
        return self._ops

    def gather(self, frame: torch.Tensor) -> torch.Tensor:
        """frame: this rank's contiguous (3,H,W) buffer whose own band rows are rendered. On return the other
        ranks' bands have been received in place (on every rank, or on `root` only). The transfers are enqueued on the
        current stream for RCCL; the call returns once they are enqueued."""
        assert frame.is_contiguous() and tuple(frame.shape) == (3, self.height, self.width)
        # (gsr_exchange_bands moves float32 words of a buffer on this rank's device: anything else would move the wrong bytes)
        assert frame.dtype == torch.float32 and frame.device == self.device, (frame.dtype, frame.device, self.device)
        if self.world == 1:
            return frame
        if self._handle is not None:
            from . import _capi
            if self._bounds_c is None:
                self.set_bounds(self.bounds)
            with torch.cuda.device(self.device):
                rc = _capi.lib().gsr_exchange_bands(self._handle, frame.data_ptr(), self.width, self.height, self._bounds_c,
                                                    -1 if self.root is None else int(self.root),
                                                    torch.cuda.current_stream(self.device).cuda_stream)
            if rc != _capi.GSR_OK:
                raise RuntimeError("gsr_exchange_bands: " + _capi.lib().gsr_exchange_last_error().decode())
            return frame
        ops = self._torch_ops(frame)
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return frame

    def rebalance(self, my_row_cost: np.ndarray, floor_cost: float = 0.0) -> list[int]:
        """my_row_cost: grid_y numbers, non-zero only for this rank's rows (instances per tile
        row). All ranks exchange them and cut the same new boundaries."""
        t = torch.as_tensor(np.asarray(my_row_cost, dtype=np.float32)).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        bounds = balanced_bands(t.cpu().numpy(), self.world, floor_cost)
        self.set_bounds(bounds)
        return bounds
