"""GPU, more than one device (skipped on a one-GPU box): the sharded frame over RCCL. One process per GPU
(torch.distributed, backend nccl); every rank renders its band of tile rows, the bands are exchanged, and the assembled
frame must equal the single-GPU frame bit for bit on every rank — for both transports of the exchange (the library's own
communicator, torch's point-to-point group), for uniform bands and for a cut that leaves a rank without rows, and for the
gather to one root."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    try:
        from gsrast_amd import camera, scenes, sharding
        from gsrast_amd.rasterizer import SplatRasterizer
        W, H = 640, 368
        scene = scenes.garden_like_scene(60_000, seed=43) if rank == 0 else None
        dev_scene = sharding.broadcast_scene(scene, device, 0)
        cam = camera.default_camera(W, H, near=0.05, far=80.0)
        r = SplatRasterizer(W, H, device=device, background=(0.1, 0.2, 0.3))
        r.configure_from_scene(dev_scene)
        whole = r.draw(cam).clone()
        grid_y = (H + 15) // 16
        notes = []
        for transport in ("rccl", "torch"):
            for root in (None, world - 1):
                ex = sharding.RowBandExchange(W, H, device, root=root, transport=transport)
                assert ex.transport == transport, ex.transport_note
                for bounds in (sharding.uniform_bands(grid_y, world), [0] + [grid_y] * world):      # (second: only rank 0 has rows)
                    ex.set_bounds(bounds)
                    r.out_color.fill_(-1.0)
                    frame = r.draw(cam, tile_rows=ex.my_tile_rows(), sync=False)
                    ex.gather(frame)
                    torch.cuda.current_stream(device).synchronize()
                    if root is None or rank == root:
                        assert torch.equal(frame, whole), (transport, root, bounds)
                ex.close()
                notes.append((transport, root))
        ret[rank] = len(notes)
    finally:
        dist.destroy_process_group()


def test_sharded_frame_over_rccl_equals_the_single_gpu_frame():
    import torch
    import torch.multiprocessing as mp
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs at least two GPUs (the one-GPU box runs the exchange's plan on the CPU and a one-rank communicator)")
    world = min(ndev, 4)                               # at most four ranks on the card(s): the box's process guard allows six
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world and all(v == 4 for v in ret.values())
