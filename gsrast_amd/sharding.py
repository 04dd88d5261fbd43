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
    no other data-path communication.
  * Balance: bands are re-cut between frames from the per-tile-row instance counts of the last
    frame (each rank knows its own rows; one small all-gather of grid_y integers).
"""
from __future__ import annotations

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
        n = torch.tensor([int(np.asarray(scene["means3D"]).shape[0])], dtype=torch.int64, device=device)
    else:
        n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    n = int(n.item())
    shapes = {"means3D": (n, 4), "scales": (n, 4), "rotations": (n, 4), "opacities": (n,), "shs": (n, 48)}
    out = {}
    for k in keys:
        if rank == src:
            t = torch.as_tensor(np.ascontiguousarray(scene[k], dtype=np.float32)).to(device)
        else:
            t = torch.empty(shapes[k], dtype=torch.float32, device=device)
        dist.broadcast(t, src)
        out[k] = t
    return out


class RowBandExchange:
    """Exchanges planar (3,H,W) row bands so that every rank ends up with the full frame.

    A band of a planar CHW frame is three contiguous pieces (rows [y0, y1) of each colour plane), so the
    exchange is one group of point-to-point transfers of exact sizes — this rank's three pieces to every
    peer, every peer's three pieces received straight into their place in the frame: no staging buffer,
    no padding to the tallest band, no copy back. On RCCL the group is one ncclGroupStart/End; each of the
    seven xGMI links of a fully connected node carries one peer's band."""

    def __init__(self, width: int, height: int, device):
        self.width, self.height = width, height
        self.grid_y = (height + TILE - 1) // TILE
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.device = device
        self.bounds = uniform_bands(self.grid_y, self.world)

    def set_bounds(self, bounds: list[int]) -> None:
        assert len(bounds) == self.world + 1 and bounds[0] == 0 and bounds[-1] == self.grid_y
        assert all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
        self.bounds = list(bounds)

    def my_tile_rows(self) -> tuple[int, int]:
        return self.bounds[self.rank], self.bounds[self.rank + 1]

    def gather(self, frame: torch.Tensor) -> torch.Tensor:
        """frame: this rank's contiguous (3,H,W) buffer whose own band rows are rendered. On return the other
        ranks' bands have been received in place. (The transfers are enqueued on the current stream for RCCL;
        the call returns once they are enqueued.)"""
        assert frame.is_contiguous() and tuple(frame.shape) == (3, self.height, self.width)
        if self.world == 1:
            return frame
        y0, y1 = band_pixel_rows(self.bounds, self.rank, self.height)
        ops = []
        for step in range(1, self.world):
            # staggered peers: in round `step` rank r sends to r + step and receives from r - step
            dst, src = (self.rank + step) % self.world, (self.rank - step) % self.world
            a, b = band_pixel_rows(self.bounds, src, self.height)
            for c in range(3):
                if y1 > y0:
                    ops.append(dist.P2POp(dist.isend, frame[c, y0:y1, :], dst))
                if b > a:
                    ops.append(dist.P2POp(dist.irecv, frame[c, a:b, :], src))
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
