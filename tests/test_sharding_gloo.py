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
