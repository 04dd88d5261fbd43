#!/usr/bin/env python3
"""Benchmark of the forward splat rasterizer (BASELINE.json metric: forward Msplats/s + fps at
1920x1080; blend-kernel HBM GB/s vs peak).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one `forward` call on one frame: allocator callbacks, preprocess, scan, the
numRendered read-back, key emission, radix sort, tile ranges, blend and the trailing device
synchronise the reference's caller performs (apps/gsrast/CudaBuffer.hpp:8-12). The scene is
resident in HBM before the timed region. With N > 1 the frame's tile rows are sharded over the
ranks and the row bands are all-gathered inside the step (strong scaling: same frame, N GPUs).

Workload at N=1: BASELINE config 2 — the Mip-NeRF360 garden .ply is not available offline, so
the "garden-like" synthetic scene of SURVEY.md §8d stands in (same splat count, labelled so).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured-achievable


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--splats", type=int, default=5_834_784)
    p.add_argument("--scene", default="garden_like", choices=["garden_like", "stress", "isotropic"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=3, help="CPU baseline renders every k-th splat (about 10 s of host time)")
    p.add_argument("--no-rebalance", action="store_true")
    p.add_argument("--backward", action="store_true",
                   help="BASELINE config 5: a step is forward + backward (gsr_backward with a fixed dL_dout); single GPU")
    p.add_argument("--overlap", action="store_true",
                   help="GSR_FLAG_OVERLAP_EMIT: block plan's emission on a second stream beside the blend (shorter frames, "
                        "but per-kernel times are then those of kernels sharing the chip)")
    p.add_argument("--plan", default="auto", choices=["auto", "sort", "blocks"], help="binning plan (GSR_FLAG_PLAN_*)")
    return p.parse_args()


def make_scene(name: str, n: int):
    from gsrast_amd import camera, scenes
    if name == "garden_like":
        sc = scenes.garden_like_scene(n, seed=43)
        span = float(np.max(sc["means3D"][:, :3].max(0) - sc["means3D"][:, :3].min(0)))
        near, far = 0.001 * span, span                 # GSRastWindow.cpp:30-36
        label = f"garden-like synthetic stand-in for the garden .ply (seed 43), N={n}"
    elif name == "stress":
        sc = scenes.stress_scene(n, seed=44)
        near, far = 0.1, 100.0
        label = f"50M-style anisotropic stress scene (seed 44), N={n}"
    else:
        sc = scenes.isotropic_scene(n, seed=42)
        near, far = 0.01, 100.0
        label = f"isotropic plumbing scene (seed 42), N={n}"
    pos = (0.0, 0.0, -25.0) if name == "stress" else (0.0, 0.0, -5.0)
    return sc, near, far, pos, label


def cpu_baseline(scene, cam, every: int):
    """Times the scalar C++ oracle (oracle/gsr_oracle.cpp, the CPU restatement of the same tile
    loop) on a bounded sample of the workload: every k-th splat, same camera and resolution."""
    from oracle import cpu_oracle
    sub = {k: np.ascontiguousarray(v[::every]) for k, v in scene.items()}
    n = int(sub["means3D"].shape[0])
    cores = cpu_oracle.hardware_concurrency() or 1
    t = {}
    st = cpu_oracle.forward(sub, cam, threads=cores, timings=t)
    return {
        "value": round(n / t["total_s"] / 1e6, 4), "unit": "Msplats/s", "cores": cores, "kind": "port",
        "sample": (f"every {every}th splat of the workload (n={n}, R={st['num_rendered']}) at the same camera and "
                   f"resolution; preprocess+sort single-thread, tile loop on {cores} std::threads; "
                   f"total {t['total_s']:.2f}s = preprocess {t['preprocess_s']:.2f} + bin/sort {t['bin_s']:.2f} + "
                   f"blend {t['blend_s']:.2f}"),
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GSR_FORCE_DIST=1 runs the sharded code path (RCCL broadcast, band all-gather, re-cut) even with one
    # rank: the only way to exercise it on a single-GPU box.
    distributed = world > 1 or os.environ.get("GSR_FORCE_DIST") == "1"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    from gsrast_amd import _capi, camera
    from gsrast_amd.rasterizer import SplatRasterizer

    W, H = args.width, args.height
    scene = None
    if rank == 0 or not distributed:
        scene, near, far, pos, label = make_scene(args.scene, args.splats)
        meta = [near, far, *pos]
    else:
        meta, label = [0.0] * 5, ""
    if distributed:
        from gsrast_amd import sharding
        m = torch.tensor(meta, dtype=torch.float64, device=device)
        dist.broadcast(m, 0)
        near, far, pos = float(m[0]), float(m[1]), tuple(float(v) for v in m[2:5])
        dev_scene = sharding.broadcast_scene(scene, device, 0)          # RCCL broadcast of the SoA
    else:
        dev_scene = scene
    cam = camera.default_camera(W, H, near=near, far=far, position=pos)

    rast = SplatRasterizer(W, H, device=device)
    rast.configure_from_scene(dev_scene)
    n_splats = rast.num_gaussians
    grid_x, grid_y = (W + 15) // 16, (H + 15) // 16
    exch = sharding.RowBandExchange(W, H, device) if distributed else None

    dl_dout = None
    if args.backward:
        assert not distributed, "--backward is a single-GPU configuration"
        dl_dout = torch.randn((3, H, W), generator=torch.Generator(device="cpu").manual_seed(7)).to(device)
    bw_ms = [0.0, 0.0]

    def step(profile=False):
        rows = exch.my_tile_rows() if exch else None
        frame = rast.draw(cam, profile=profile, tile_rows=rows, sync=not distributed, plan=args.plan,
                          overlap_emit=args.overlap)
        if dl_dout is not None:
            rast.backward(dl_dout, profile=profile)
            if profile:
                bw_ms[0] += rast.last_backward_ms[0]
                bw_ms[1] += rast.last_backward_ms[1]
        if exch:
            exch.gather(frame)
            torch.cuda.current_stream(device).synchronize()
            _capi.check(rast.lib.gsr_poll_async_error(), "gsr_forward (device side)")
        return frame

    def sync_all():
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(device)

    # warm-up (also converges the row-band balance when sharded)
    for w in range(args.warmup):
        step()
        if exch and not args.no_rebalance and w < args.warmup - 1:
            r = rast.map_image_state()["ranges"].to(torch.int64)
            per_tile = (r[:, 1] - r[:, 0]).clamp(min=0).view(grid_y, grid_x).sum(1).cpu().numpy().astype(np.float64)
            b0, b1 = exch.my_tile_rows()
            mine = np.zeros(grid_y)
            mine[b0:b1] = per_tile[b0:b1]
            exch.rebalance(mine, floor_cost=0.02 * float(per_tile[b0:b1].mean() if b1 > b0 else 0.0) + 1.0)

    stage_sum = {}
    staged_frame = rast.draw(cam, count_staged=True, tile_rows=exch.my_tile_rows() if exch else None, plan=args.plan)
    plan_used = rast.last_plan
    r_f = rast.last_records_staged
    num_rendered = rast.last_num_rendered
    geo = rast.map_geometry_state()
    n_visible = int((geo["tilesTouched"] != 0).sum().item())        # V of this rank's band
    n_visible_total = int((geo["radii"] > 0).sum().item())

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    # Per-stage device times come from extra, untimed frames with GSR_FLAG_PROFILE (HIP events around
    # every stage cost a few tens of microseconds per frame, which the timed frames do not pay).
    prof_steps = max(5, min(args.steps, 20))
    bw_ms[0] = bw_ms[1] = 0.0
    t1 = time.perf_counter()
    for _ in range(prof_steps):
        step(profile=True)
        for k, v in rast.last_stage_ms.items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
    sync_all()
    elapsed_profiled = time.perf_counter() - t1
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([float(r_f), float(num_rendered)], dtype=torch.float64, device=device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        r_f_total, r_total = int(cnt[0].item()), int(cnt[1].item())
    else:
        r_f_total, r_total = r_f, num_rendered

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        stage_ms = {k: v / prof_steps for k, v in stage_sum.items()}
        # ALGORITHMIC bytes per launch on THIS rank (SURVEY.md §8d / BASELINE.md §2). N splats,
        # V visible, R instances, R_f records staged by the blend, P pixels, T tiles of this rank.
        rows = exch.my_tile_rows() if exch else (0, grid_y)
        px_rows = min(rows[1] * 16, H) - min(rows[0] * 16, H)
        P_loc, T_loc = px_rows * W, (rows[1] - rows[0]) * grid_x
        N, R = n_splats, num_rendered
        blocks = plan_used == "blocks"
        alg = {
            "preprocess": N * (52 + 12 + 8) + n_visible * 72,
            "scan": 8 * N,
            # sort plan: the emission kernel of SURVEY.md §8d. block plan: block_emit_kernel writes the
            # SORTED pairs once (12 B each); its block-list reads (12 B per entry, E <= R) are not counted.
            "duplicate": 12 * R if blocks else 8 * N + 20 * n_visible + 12 * R,
            "sort_pass1": 0 if blocks else 24 * R,   # one onesweep launch: 12 B read + 12 B written per pair
            "sort_pass2": 24 * R,
            "ranges": 0 if blocks else 16 * T_loc * max(1, int(np.ceil(np.log2(max(R, 2))))) + 8 * T_loc,   # two binary searches per tile
            "blend": 40 * r_f + 20 * P_loc + 8 * T_loc,
        }
        kernels = {}
        for k, b in alg.items():
            t = stage_ms.get(k, 0.0)
            if t > 0:
                gbs = b / (t * 1e-3) / 1e9
                kernels[k] = {"ms": round(t, 4), "alg_bytes": int(b), "gbs": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
        dom = max(("sort_pass1", "sort_pass2", "blend", "duplicate", "preprocess", "ranges"),
                  key=lambda k: stage_ms.get(k, 0.0))
        dom_names = {"sort_pass1": "onesweep_kernel<u64> (tile-column digit pass)",
                     "sort_pass2": "onesweep_kernel<u64> (tile-row digit pass)",
                     "blend": "blend_blocks_kernel" if blocks else "blend_wave_kernel",
                     "duplicate": "block_emit_kernel (sorted lists written directly)" if blocks else "emit_chunk_kernel", "preprocess": "preprocess_kernel", "ranges": "tile_ranges_kernel"}

        # HBM bytes per launch measured with PMC counters in separate rocprofv3 passes of this same
        # command (profiles/): valid only for the default single-GPU workload they were taken on.
        traffic = {}
        default_frame = (not distributed and args.scene == "garden_like" and args.splats == 5_834_784 and (W, H) == (1920, 1080))
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_r01.json")
        if default_frame and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("blocks" if blocks else "sort", {})

        def roof(k, note):
            e = kernels.get(k, {"gbs": 0.0, "ms": 0.0, "alg_bytes": 0})
            return {"bound": "hbm", "kernel": dom_names[k], "achieved": e["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(e["gbs"] / HBM_PEAK_GBS, 5), "traffic": traffic.get(k),
                    "traffic_source": "profiles/pmc_traffic_r01.json (rocprofv3 --pmc, separate passes)" if k in traffic else None,
                    "algorithmic_bytes_per_launch": e["alg_bytes"], "avg_launch_ms": e["ms"], "note": note}

        if args.backward:
            stage_ms["render_backward"] = bw_ms[0] / prof_steps
            stage_ms["chain_backward"] = bw_ms[1] / prof_steps
        out = {
            "metric": "forward_backward_msplats_per_s" if args.backward else "forward_msplats_per_s",
            "value": round(n_splats / (ms_per_step * 1e-3) / 1e6, 3),
            "unit": "Msplats/s",
            "fps": round(1e3 / ms_per_step, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{label}, {W}x{H} forward, fixed reference default camera",
                       "width": W, "height": H, "splats": n_splats, "visible": n_visible_total, "num_rendered": r_total,
                       "records_staged": r_f_total, "minstances_per_s": round(r_total / (ms_per_step * 1e-3) / 1e6, 2),
                       "parallelism": f"tile-rows x{world}" if distributed else "single GPU", "binning_plan": plan_used,
                       "bands": exch.bounds if exch else None},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_meaning": ({"depth_order": "visible-key compaction + depth sort + block lists", "sort_pass1": "unit masks + prefixes + tile ranges",
                                  "duplicate": "block_emit_kernel (the sorted keys / values)", "blend": "blend_blocks_kernel"} if blocks else
                                 {"depth_order": "visible-key compaction + depth sort + column counts / scan", "duplicate": "emit_chunk_kernel",
                                  "sort_pass2": "onesweep pass on the tile row", "blend": "blend_wave_kernel"}),
            "stage_ms_source": (f"HIP events recorded by the library on the launching stream (GSR_FLAG_PROFILE) over {prof_steps} "
                                f"frames of the same workload run right after the timed region; those frames took "
                                f"{elapsed_profiled / prof_steps * 1e3:.4f} ms each with the events in place"),
            "roofline": roof(dom, "dominant kernel of this frame; HIP-event time of the launch on its own stream"),
            "roofline_blend": roof("blend", "the kernel BASELINE.json names; VALU-bound (~100 flop/B), fraction as measured"),
            "kernels": kernels,
        }
        if not distributed and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, cam, args.cpu_sample)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)          # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()
