"""CPU, world_size 2 and 3 over gloo: the multi-GPU path's collectives — scene broadcast, row-band
all-gather into a full frame, cost exchange + re-cut of the bands — with the renderer replaced
by slices of a precomputed oracle frame (tests may use the oracle as the checker)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gsrast_amd import camera, scenes, sharding
        from oracle import cpu_oracle
        dev = torch.device("cpu")
        W, H = 200, 120
        scene = scenes.isotropic_scene(800, 42) if rank == 0 else None
        got = sharding.broadcast_scene(scene, dev, 0)
        ref_scene = scenes.isotropic_scene(800, 42)
        for k, v in got.items():
            assert np.array_equal(v.numpy(), ref_scene[k]), k
        cam = camera.default_camera(W, H)
        full = torch.from_numpy(cpu_oracle.forward(ref_scene, cam, (0.1, 0.2, 0.3))["out_color"])
        ex = sharding.RowBandExchange(W, H, dev)
        for trial in range(2):
            b0, b1 = ex.my_tile_rows()
            y0, y1 = min(b0 * 16, H), min(b1 * 16, H)
            local = torch.full((3, H, W), -7.0)                     # only the own band is "rendered"
            local[:, y0:y1, :] = full[:, y0:y1, :]
            frame = ex.gather(local)
            assert torch.equal(frame, full), f"trial {trial}: assembled frame differs"
            cost = np.zeros(ex.grid_y)
            cost[b0:b1] = np.arange(b0, b1) ** 2 + 1.0               # skewed: later rows are heavier
            bounds = ex.rebalance(cost)
            assert bounds[0] == 0 and bounds[-1] == ex.grid_y
        ret[rank] = tuple(ex.bounds)
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_band_exchange_and_rebalance(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] == ret[0] for r in range(world))               # every rank cuts the same bands
    assert ret[0][1] > 8 // world                                             # boundary moved towards the heavy rows


def _worker_modes(rank, world, port, ret):
    """Gather to one root, bands that are empty on some ranks, the op list kept between frames, and the host time of a call."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gsrast_amd import sharding
        dev = torch.device("cpu")
        W, H = 96, 200                                   # 13 tile rows, the last one 8 pixel rows high
        ys = torch.arange(H, dtype=torch.float32)[None, :, None]
        want = (torch.arange(3, dtype=torch.float32)[:, None, None] * 4096.0 + ys).expand(3, H, W).contiguous()
        grid_y = (H + 15) // 16
        empty_some = [0] + [min(grid_y, 5 * g) for g in range(1, world)] + [grid_y]        # ranks past the third get no rows
        for root in (None, world - 1):
            ex = sharding.RowBandExchange(W, H, dev, root=root)
            assert ex.transport == "torch"               # (gloo: the library's RCCL path is not even tried)
            frame = torch.empty((3, H, W))
            for bounds in (sharding.uniform_bands(grid_y, world), empty_some, sharding.uniform_bands(grid_y, world)):
                ex.set_bounds(bounds)
                for _ in range(2):                       # the second call reuses the op list
                    y0, y1 = sharding.band_pixel_rows(ex.bounds, rank, H)
                    frame.fill_(-3.0)
                    frame[:, y0:y1, :] = want[:, y0:y1, :]
                    ops_before = ex._ops
                    ex.gather(frame)
                    if root is None or rank == root:
                        assert torch.equal(frame, want), (root, bounds)
                    else:                                # a non-root rank receives nothing: only its own band is there
                        assert bool((frame[:, :y0, :] == -3.0).all()) and bool((frame[:, y1:, :] == -3.0).all())
                assert ex._ops is ops_before             # same frame buffer, same bands: not rebuilt
            dist.barrier()
        # host time of one call (gloo moves the bytes on the calling thread, so this is an upper bound of what the
        # enqueue costs on RCCL; reported, not asserted)
        ex = sharding.RowBandExchange(W, H, dev)
        frame = want.clone()
        for _ in range(3):
            ex.gather(frame)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            ex.gather(frame)
        ret[rank] = (time.perf_counter() - t0) / 20 * 1e6
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gather_to_root_empty_bands_and_kept_op_list(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_modes, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    print(f"RowBandExchange.gather over gloo, world {world}: {max(ret.values()):.0f} us per call on the slowest rank")


def test_the_librarys_exchange_plan_pairs_up_across_ranks():
    """gsr_exchange_plan (host only) lists the ncclSend / ncclRecv calls gsr_exchange_bands issues. For every world size,
    uneven / empty bands and both modes: each send has the receive that RCCL will match it with (same pair of ranks, same
    position in that pair's sequence, same count, the receiver's offset = the sender's: a band lands where it came from),
    and the receives of a rank cover exactly the rows it does not own."""
    import ctypes as C
    from gsrast_amd import _capi
    L = _capi.lib()
    W, H = 100, 1080
    grid_y = (H + 15) // 16
    rng = np.random.default_rng(1)

    def plan(rank, world, bounds, root):
        n_max = 6 * world
        arr = (C.c_int32 * (world + 1))(*bounds)
        s, p = (C.c_int32 * n_max)(), (C.c_int32 * n_max)()
        off, cnt = (C.c_uint64 * n_max)(), (C.c_uint64 * n_max)()
        n = L.gsr_exchange_plan(rank, world, W, H, arr, root, n_max, s, p, off, cnt)
        assert 0 <= n <= n_max
        return [(bool(s[i]), int(p[i]), int(off[i]), int(cnt[i])) for i in range(n)]

    for world in (1, 2, 3, 5, 8):
        cuts = [[(grid_y * g) // world for g in range(world + 1)]]
        for _ in range(3):
            inner = sorted(int(v) for v in rng.integers(0, grid_y + 1, world - 1))
            cuts.append([0] + inner + [grid_y])
        for bounds in cuts:
            for root in (-1, world - 1):
                plans = [plan(r, world, bounds, root) for r in range(world)]
                for r in range(world):
                    for q in range(world):
                        sends = [(o, c) for (s, p, o, c) in plans[r] if s and p == q]
                        recvs = [(o, c) for (s, p, o, c) in plans[q] if not s and p == r]
                        assert sends == recvs, (world, bounds, root, r, q)
                    covered = np.zeros(3 * W * H, bool)
                    for (s, p, o, c) in plans[r]:
                        if not s:
                            assert not covered[o:o + c].any()
                            covered[o:o + c] = True
                    own = np.zeros((3, H, W), bool)
                    own[:, min(bounds[r] * 16, H):min(bounds[r + 1] * 16, H), :] = True
                    if root < 0 or r == root:
                        assert np.array_equal(covered, ~own.reshape(-1)), (world, bounds, root, r)
                    else:
                        assert not covered.any()
    # malformed bands are refused
    bad = (C.c_int32 * 3)(0, 70, 68)
    assert L.gsr_exchange_plan(0, 2, W, H, bad, -1, 0, None, None, None, None) == -_capi.GSR_ERR_INVALID_ARG
