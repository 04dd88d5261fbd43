#!/usr/bin/env python3
"""Benchmark of the forward splat rasterizer (BASELINE.json metric: forward Msplats/s + fps at
1920x1080; blend-kernel HBM GB/s vs peak).

  python bench.py --gpus N --steps K --warmup W
      N = 1: runs in this process. N > 1 without a launcher around it: starts
      `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD process (before
      anything here touches the GPU) and relays its JSON line; fails loudly if fewer than N devices are visible.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
      what the driver runs: one rank per GPU over RCCL (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).

A step is one `forward` call on one frame: allocator callbacks, preprocess, scan, the numRendered read-back, binning,
blend and the trailing device synchronise the reference's caller performs (apps/gsrast/CudaBuffer.hpp:8-12). The
scene is resident in HBM before the timed region. With N > 1 the frame's tile rows are sharded over the ranks and the
row bands are exchanged inside the step (strong scaling: same frame, N GPUs); BASELINE config 3 (the same scene at
3840 x 2160) is then timed as well and reported under "config3_4k".

Workload at N=1: BASELINE config 2 — the Mip-NeRF360 garden .ply is not available offline, so the "garden-like"
synthetic scene of SURVEY.md §8d stands in (same splat count, labelled so), at the reference's default pose. Extra keys
at N=1 (measured after, and outside, the timed region of the headline): the same scene from a pose outside the cloud,
and from far away, a blend-bound variant (opacities x 0.1) and the forward-only mode without the sorted lists.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# Multi-process GPU work on this pool needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise): set before the HIP
# runtime comes up, for every way this file is started — the driver's own torch.distributed.run included.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured-achievable
N_SIMD = 1024              # 256 CUs x 4 SIMDs
DEFAULT_SPLATS = 5_834_784


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--spinup", type=int, default=16,
                   help="untimed frames rendered before anything is measured (before the W warm-up steps): the chip's clocks take "
                        "about ten frames to come back after the idle milliseconds of allocation and set-up (scripts/clock_ramp.py, "
                        "profiles/r05_clock_ramp.txt); reported as config.spinup_frames; 0 = none")
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--splats", type=int, default=DEFAULT_SPLATS)
    p.add_argument("--scene", default="garden_like", choices=["garden_like", "stress", "isotropic"])
    p.add_argument("--ply", default=os.environ.get("GSR_GARDEN_PLY", ""),
                   help="a trained 3DGS .ply in the reference's format (apps/gsrast/SplatData.cpp:114-156), e.g. Mip-NeRF360 garden: "
                        "replaces the synthetic stand-in when the file exists (default: $GSR_GARDEN_PLY); loaded through "
                        "gsr_ply_activate, camera as the app sets it up (GSRastWindow.cpp:30-36)")
    p.add_argument("--pose", default=None, help="camera position x,y,z (default: the reference's (0,0,-5); (0,0,-25) for --scene stress)")
    p.add_argument("--opacity-scale", type=float, default=1.0, help="multiplies every opacity (0.1: blend-bound variant, R_f ~ R)")
    p.add_argument("--semantics", default="gscuda", choices=["gscuda", "inria"],
                   help="inria: the upstream rasterizer's semantics (GSR_FLAG_SEMANTICS_INRIA), SH evaluated up to --sh-degree")
    p.add_argument("--sh-degree", type=int, default=3)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=1,
                   help="CPU baseline renders every k-th splat of the workload (1 = the whole frame: 20-30 s on the box's host cores, "
                        "about 14 GB of host memory for the 267 M pairs of the default frame)")
    p.add_argument("--no-rebalance", action="store_true")
    p.add_argument("--gather", default="all", choices=["all", "root"],
                   help="N > 1: every rank receives every band (all), or only rank 0 does (root: 1/N of the traffic when one rank displays)")
    p.add_argument("--no-extras", action="store_true", help="skip the extra frames (second pose, blend-bound, no sorted lists, config 3 at N > 1)")
    p.add_argument("--backward", action="store_true",
                   help="BASELINE config 5: a step is forward + backward (gsr_backward with a fixed dL_dout); single GPU")
    p.add_argument("--overlap", action="store_true",
                   help="GSR_FLAG_OVERLAP_EMIT: force the block plan's blend onto a second stream beside the emission (default: the library decides per call)")
    p.add_argument("--serial-emit", action="store_true",
                   help="GSR_FLAG_SERIAL_EMIT: never beside each other (per-kernel times are then those of the kernels alone)")
    p.add_argument("--no-sorted-lists", action="store_true", help="GSR_FLAG_NO_SORTED_LISTS for the headline frame (forward-only callers)")
    p.add_argument("--colors-precomp", action="store_true",
                   help="pass the colours as the reference's colorsPrecomp argument (GSCuda.cuh:111), computed once per scene by "
                        "gsr_colors_from_dc: the preprocess then reads no SH and writes no geomState.rgb (gscuda semantics)")
    p.add_argument("--plan", default="auto", choices=["auto", "sort", "blocks"], help="binning plan (GSR_FLAG_PLAN_*)")
    p.add_argument("--no-tile-history", action="store_true", help="GSR_FLAG_NO_TILE_HISTORY for the headline frame")
    p.add_argument("--no-deep-tiles", action="store_true", help="GSR_FLAG_NO_DEEP_TILES for the headline frame: one wave per tile whatever the instances per Gaussian")
    p.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched ranks (0: a free one)")
    p.add_argument("--dry-run", action="store_true",
                   help="launcher / rendezvous / band-exchange plumbing check on CPU over gloo: renders nothing, measures "
                        "nothing, prints a line with value null")
    return p.parse_args()


# ---- launching N ranks ------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    """--gpus N > 1 and no launcher around us: run the ranks as a child `torch.distributed.run` and relay the JSON
    line. Nothing in this process touches the GPU (device_count does not initialise it on this image), and the
    current process is never replaced."""
    if not args.dry_run:
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} asked for, but {ndev} HIP device(s) visible: refusing to report a "
                  f"{args.gpus}-GPU number from fewer GPUs", file=sys.stderr)
            return 2
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = child.stdout.splitlines()
    json_lines = [ln for ln in lines if ln.startswith("{") and ln.rstrip().endswith("}")]
    for ln in lines:
        if not json_lines or ln is not json_lines[-1]:
            print(ln, file=sys.stderr)
    if child.returncode != 0 or not json_lines:
        print(f"bench.py: the {args.gpus}-rank child exited with code {child.returncode}"
              + ("" if json_lines else " without a JSON line"), file=sys.stderr)
        return child.returncode or 3
    print(json_lines[-1], flush=True)
    return 0


def dry_run(args) -> int:
    """CPU / gloo: every rank fills its band of a synthetic frame, the bands are exchanged, every rank checks the
    assembled frame, the bands are re-cut once and exchanged again. No rasterizer, no number."""
    import torch.distributed as dist
    from gsrast_amd import sharding
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = args.width, args.height
    dev = torch.device("cpu")
    ex = sharding.RowBandExchange(W, H, dev)
    ys = torch.arange(H, dtype=torch.float32)[None, :, None]
    want = (torch.arange(3, dtype=torch.float32)[:, None, None] * 1000.0 + ys).expand(3, H, W).contiguous()
    ok = True
    bands = []
    for trial in range(2):
        b0, b1 = ex.my_tile_rows()
        y0, y1 = min(b0 * 16, H), min(b1 * 16, H)
        frame = torch.full((3, H, W), -1.0)
        frame[:, y0:y1, :] = want[:, y0:y1, :]
        ex.gather(frame)
        ok = ok and bool(torch.equal(frame, want))
        bands.append(list(ex.bounds))
        cost = np.zeros(ex.grid_y)
        cost[b0:b1] = np.arange(b0, b1) + 1.0
        ex.rebalance(cost)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    n = dist.get_world_size()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "forward_msplats_per_s", "value": None, "unit": "Msplats/s", "n_gpus": n, "dry_run": True,
                          "exchange_ok": bool(flag.item() == 1.0), "bands": bands,
                          "note": "CPU / gloo plumbing check of the launcher and the band exchange: nothing rendered or measured"}),
              flush=True)
    return 0 if flag.item() == 1.0 else 4


# ---- workload ---------------------------------------------------------------------------------------------------------
def make_scene(name: str, n: int, device, full_sh: bool = False, ply_path: str = ""):
    """Returns (scene dict, near, far, default position, label). The 50 M stress scene is generated on the device (same
    splitmix64 definition, gsrast_amd/scenes.py); the others on the host as in round 1. ply_path: a scene file in the
    reference's format instead (its label starts with "ply:")."""
    from gsrast_amd import scenes
    if ply_path:
        from gsrast_amd import ply
        sc = ply.load_ply(ply_path, device=device, sh_layout="coefficient_major" if full_sh else "file")
        span = float((sc["bbox_max"] - sc["bbox_min"]).max().item())      # GSRastWindow.cpp:30-36: far = largest bbox span
        sc = {k: sc[k] for k in ("means3D", "scales", "rotations", "opacities", "shs")}
        n_file = int(sc["means3D"].shape[0])
        return sc, 0.001 * span, span, (0.0, 0.0, -5.0), f"ply:{os.path.basename(ply_path)} (N={n_file}, the reference's loader semantics)"
    if name == "garden_like":
        sc = scenes.garden_like_scene(n, seed=43)
        span = float(np.max(sc["means3D"][:, :3].max(0) - sc["means3D"][:, :3].min(0)))
        near, far = 0.001 * span, span                 # GSRastWindow.cpp:30-36
        label = f"garden-like synthetic stand-in for the garden .ply (seed 43), N={n}"
        pos = (0.0, 0.0, -5.0)
    elif name == "stress":
        sc = scenes.stress_scene_device(n, seed=44, device=device, full_sh=full_sh)
        near, far = 0.1, 100.0
        label = f"50M-style anisotropic stress scene (seed 44{', 48 SH floats' if full_sh else ''}), N={n}"
        pos = (0.0, 0.0, -25.0)
    else:
        sc = scenes.isotropic_scene(n, seed=42)
        near, far = 0.01, 100.0
        label = f"isotropic plumbing scene (seed 42), N={n}"
        pos = (0.0, 0.0, -5.0)
    if full_sh and name != "stress":
        rng = np.random.default_rng(45)
        sc["shs"][:, 3:] = rng.normal(0.0, 0.3, (n, 45)).astype(np.float32)
    return sc, near, far, pos, label


def cpu_baseline(scene, cam, every: int):
    """Times the scalar C++ oracle (oracle/gsr_oracle.cpp, the CPU restatement of the same pipeline: preprocess, key
    duplication, stable 64-bit sort, tile ranges, tile loop) on the bench workload itself (every == 1) or on every k-th
    splat of it, same camera and resolution, on all host cores."""
    from gsrast_amd import scenes
    from oracle import cpu_oracle
    sub = scenes.scene_rows(scene, slice(None, None, every))        # (host float32 copies, whatever the scene is held as)
    n = int(sub["means3D"].shape[0])
    cores = cpu_oracle.hardware_concurrency() or 1
    t = {}
    st = cpu_oracle.forward(sub, cam, threads=cores, timings=t)
    whole = every == 1
    return {
        "value": round(n / t["total_s"] / 1e6, 4), "unit": "Msplats/s", "cores": cores, "kind": "port",
        "comparable": whole,        # True: the same frame as `value` (same N, same R); False: a sub-sampled frame, fewer instances per splat
        "sample": ((f"the bench frame itself (n={n}, R={st['num_rendered']})" if whole else
                    f"every {every}th splat of the workload (n={n}, R={st['num_rendered']}: NOT the bench frame)")
                   + f" at the same camera and resolution; preprocess, key duplication, stable sort and tile loop on {cores} "
                     f"std::threads (scan and tile ranges: one thread); total {t['total_s']:.2f}s = preprocess {t['preprocess_s']:.2f} + "
                     f"bin/sort {t['bin_s']:.2f} + blend {t['blend_s']:.2f}; the key duplication + index-indirect std::stable_sort of the R "
                     f"pairs is {t['bin_s'] / max(t['total_s'], 1e-9):.0%} of it (the tile loop BASELINE names: {t['blend_s']:.2f}s)"),
        "sort_share": round(t["bin_s"] / max(t["total_s"], 1e-9), 4),
    }


class Runner:
    """One rasterizer + (when sharded) its band exchange; `measure` times K steps of it."""

    def __init__(self, dev_scene, W, H, device, distributed, args):
        from gsrast_amd import _capi, sharding
        from gsrast_amd.rasterizer import SplatRasterizer
        self.capi = _capi
        self.W, self.H, self.device, self.distributed, self.args = W, H, device, distributed, args
        self.rast = SplatRasterizer(W, H, device=device)
        self.rast.configure_from_scene(dev_scene)
        self.exch = sharding.RowBandExchange(W, H, device, root=0 if args.gather == "root" else None) if distributed else None
        self.grid_x, self.grid_y = (W + 15) // 16, (H + 15) // 16
        self.dl_dout = None
        self.bw_ms = [0.0, 0.0]
        self.bw_outputs = None              # gsr_backward's optional outputs: None = every array
        self.split_ms = [0.0, 0.0]          # sharded runs: device time of the render and of the band exchange (HIP events, profiled frames)

    def step(self, cam, profile=False, **kw):
        rast, exch = self.rast, self.exch
        rows = exch.my_tile_rows() if exch else None
        timed = profile and exch is not None          # (the profiled frames after the timed region: render / exchange split per rank)
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        frame = rast.draw(cam, profile=profile, tile_rows=rows, sync=not self.distributed, **kw)
        if timed:
            ev[1].record()
        if self.dl_dout is not None:
            rast.backward(self.dl_dout, profile=profile, semantics=kw.get("semantics", "gscuda"), sh_degree=kw.get("sh_degree", 3),
                          wide_sums=os.environ.get("GSR_BW_WIDE_SUMS", "1") != "0", outputs=self.bw_outputs)
            if profile:
                self.bw_ms[0] += rast.last_backward_ms[0]
                self.bw_ms[1] += rast.last_backward_ms[1]
        if exch:
            exch.gather(frame)
            if timed:
                ev[2].record()
            torch.cuda.current_stream(self.device).synchronize()
            rast.poll_async_error()
            if timed:
                self.split_ms[0] += ev[0].elapsed_time(ev[1])
                self.split_ms[1] += ev[1].elapsed_time(ev[2])
        return frame

    def sync_all(self):
        torch.cuda.synchronize(self.device)
        if self.distributed:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(self.device)

    def measure(self, cam, steps, warmup, **kw):
        rast, exch = self.rast, self.exch

        def counted_frame():
            # (one untimed frame with GSR_FLAG_COUNT_STAGED and the reads of the frame's figures)
            rast.draw(cam, count_staged=True, tile_rows=exch.my_tile_rows() if exch else None, **kw)
            o = {"plan": rast.last_plan, "lists_written": rast.last_lists_written, "blend_from_lists": rast.last_blend_from_lists,
                 "records_staged": rast.last_records_staged, "num_rendered": rast.last_num_rendered}
            geo = rast.map_geometry_state()
            o["visible_band"] = int((geo["tilesTouched"] != 0).sum().item())
            o["visible"] = int((geo["radii"] > 0).sum().item())
            return o

        # The chip's clocks take about ten frames (12 ms) to come back after a few milliseconds of idling — allocations, the
        # scene's upload, a host-side read — whatever the library does (profiles/r05_clock_ramp.txt): `spinup` untimed frames
        # bring them up before anything is measured (config.spinup_frames says how many), and on a single GPU the frame's
        # figures are read BEFORE the W warm-up steps, so that those run straight into the K timed ones. Sharded: after
        # them — the warm-up is what settles the bands the figures describe.
        spin = max(0, int(getattr(self.args, "spinup", 0)))
        for _ in range(spin):
            self.step(cam, **kw)
        out = counted_frame() if not exch else None
        for w in range(warmup):                     # warm-up (also converges the row-band balance when sharded)
            self.step(cam, **kw)
            if exch and not self.args.no_rebalance and w < warmup - 1:
                r = rast.map_image_state()["ranges"].to(torch.int64)
                per_tile = (r[:, 1] - r[:, 0]).clamp(min=0).view(self.grid_y, self.grid_x).sum(1).cpu().numpy().astype(np.float64)
                b0, b1 = exch.my_tile_rows()
                mine = np.zeros(self.grid_y)
                mine[b0:b1] = per_tile[b0:b1]
                exch.rebalance(mine, floor_cost=0.02 * float(per_tile[b0:b1].mean() if b1 > b0 else 0.0) + 1.0)
        if out is None:
            out = counted_frame()
        out["untimed_frames"] = spin + 1 + warmup
        self.sync_all()
        t0 = time.perf_counter()
        stamps = [t0]
        for _ in range(steps):
            self.step(cam, **kw)                    # (ends with the stream's synchronise: a step's work is complete when it returns)
            stamps.append(time.perf_counter())
        self.sync_all()
        out["elapsed"] = time.perf_counter() - t0
        per_step = np.diff(np.asarray(stamps)) * 1e3
        out["ms_median"], out["ms_p99"] = float(np.median(per_step)), float(np.percentile(per_step, 99))
        out["tiles_reordered"] = bool(rast.last_tiles_reordered)
        out["deep_tiles"] = bool(getattr(rast, "last_deep_tiles", False))
        out["emit_overlapped"] = bool(rast.last_emit_overlapped)
        out["colors_beside"] = bool(rast.last_colors_beside)
        # Per-stage device times come from extra, untimed frames with GSR_FLAG_PROFILE (HIP events around
        # every stage cost a few tens of microseconds per frame, which the timed frames do not pay).
        prof_steps = max(5, min(steps, 20))
        stage_sum = {}
        self.bw_ms = [0.0, 0.0]
        self.split_ms = [0.0, 0.0]
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            self.step(cam, profile=True, **kw)
            for k, v in rast.last_stage_ms.items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
        self.sync_all()
        out["profiled_ms_per_step"] = (time.perf_counter() - t1) / prof_steps * 1e3
        out["stage_ms"] = {k: v / prof_steps for k, v in stage_sum.items()}
        out["prof_steps"] = prof_steps
        out["backward_ms"] = [v / prof_steps for v in self.bw_ms]
        out["tile_rows"] = list(exch.my_tile_rows()) if exch else [0, self.grid_y]
        out["bands"] = list(exch.bounds) if exch else None
        if self.distributed:
            import torch.distributed as dist
            t = torch.tensor([out["elapsed"]], dtype=torch.float64, device=self.device)
            mine = torch.tensor([out["elapsed"], float(out["records_staged"]), float(out["num_rendered"]),
                                 float(out["tile_rows"][0]), float(out["tile_rows"][1]),
                                 self.split_ms[0] / prof_steps, self.split_ms[1] / prof_steps], dtype=torch.float64, device=self.device)
            allr = torch.zeros((dist.get_world_size(), 7), dtype=torch.float64, device=self.device)
            dist.all_gather_into_tensor(allr.view(-1), mine)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out["elapsed"] = float(t.item())                       # MAX over ranks
            allr = allr.cpu().numpy()
            out["per_rank"] = [{"rank": g, "tile_rows": [int(allr[g, 3]), int(allr[g, 4])], "num_rendered": int(allr[g, 2]),
                                "records_staged": int(allr[g, 1]), "ms_per_step": round(allr[g, 0] / steps * 1e3, 4),
                                # device time between the first launch of the frame and its last kernel / the end of the band
                                # exchange behind it (HIP events on the launching stream, profiled frames)
                                "render_ms": round(float(allr[g, 5]), 4), "exchange_ms": round(float(allr[g, 6]), 4)}
                               for g in range(allr.shape[0])]
            out["records_staged_total"], out["num_rendered_total"] = int(allr[:, 1].sum()), int(allr[:, 2].sum())
        else:
            out["per_rank"] = None
            out["records_staged_total"], out["num_rendered_total"] = out["records_staged"], out["num_rendered"]
        out["ms_per_step"] = out["elapsed"] / steps * 1e3
        return out


def walk_path(W, H, near, far, frames=120):
    """A viewer's frames (FirstPersonCamera.cpp:28-38, Inspector.cpp:60-62: the reference's only metric is the frame rate
    of a moving first-person camera): the camera walks from (0,0,-30), outside the cloud, to the reference's default pose
    (0,0,-5) inside it — two thirds of the frames, crossing the instances-per-Gaussian thresholds of the binning plan and
    of the blend's feed — and then turns on the spot (yaw to 0.9 rad, a little pitch)."""
    import math
    from gsrast_amd import camera
    cams = []
    walk = (2 * frames) // 3
    for i in range(frames):
        if i < walk:
            z, yaw, pitch = -30.0 + 25.0 * i / (walk - 1), 0.0, 0.0
        else:
            t = (i - walk + 1) / (frames - walk)
            z, yaw, pitch = -5.0, 0.9 * t, 0.15 * math.sin(3.0 * t)
        cams.append(camera.first_person_camera((0.0, 0.0, z), yaw, pitch, math.radians(45.0), near, far, W, H, True))
    return cams


def random_views(W, H, near, far, frames=60, seed=5):
    """A trainer's frames (BASELINE config 5's access pattern): unrelated cameras of one size, every call another — eyes
    between 3 and 30 units from the origin in any direction, each looking at a point near the origin."""
    import math
    from gsrast_amd import camera
    rng = np.random.default_rng(seed)
    cams = []
    for _ in range(frames):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        eye = d * rng.uniform(3.0, 30.0)
        front = rng.normal(scale=1.0, size=3) - eye
        front /= np.linalg.norm(front)
        yaw, pitch = math.atan2(front[0], front[2]), math.asin(max(-1.0, min(1.0, front[1])))
        cams.append(camera.first_person_camera(tuple(eye), yaw, pitch, math.radians(45.0), near, far, W, H, True))
    return cams


def time_sequence(run, cams, draw_kw, backward=False, repeats=3):
    """Every camera of the sequence rendered ONCE per pass, a device synchronise after each frame (what the reference's
    caller does); one untimed pass first (the chunks grow to the sequence's largest frame), then `repeats` timed passes
    with the tile history and as many without, alternating. Host wall time per frame."""
    rast = run.rast
    saved = run.dl_dout
    if backward and run.dl_dout is None:
        run.dl_dout = torch.randn((3, run.H, run.W), generator=torch.Generator(device="cpu").manual_seed(7)).to(run.device)
    if not backward:
        run.dl_dout = None

    def one_pass(hist, record):
        times, plans, overl, dropped, reordered = [], [], 0, 0, 0
        for cam in cams:
            t0 = time.perf_counter()
            run.step(cam, **{**draw_kw, "tile_history": hist})
            times.append((time.perf_counter() - t0) * 1e3)
            plans.append(rast.last_plan + ("/lists" if rast.last_blend_from_lists else ""))
            overl += int(rast.last_emit_overlapped)
            dropped += int(rast.last_tile_order_dropped)
            reordered += int(rast.last_tiles_reordered)
        if record is not None:
            record.append({"ms": np.asarray(times), "flips": sum(1 for a, b in zip(plans, plans[1:]) if a != b),
                           "overlapped": overl, "dropped": dropped, "reordered": reordered})
    one_pass(True, None)             # (untimed: the chunks grow, and every kernel either mode uses has been loaded once)
    one_pass(False, None)
    with_h, without = [], []
    for _ in range(repeats):
        one_pass(True, with_h)
        one_pass(False, without)
    run.dl_dout = saved
    allw, allo = np.concatenate([r["ms"] for r in with_h]), np.concatenate([r["ms"] for r in without])
    return {"frames": len(cams), "passes": repeats,
            "hist_over_nohist": round(float(allw.sum() / allo.sum()), 4),      # (the same frames, summed: below 1 = the history pays)
            "ms_mean": round(float(allw.mean()), 4), "ms_p50": round(float(np.median(allw)), 4), "ms_p99": round(float(np.percentile(allw, 99)), 4),
            "nohist_ms_mean": round(float(allo.mean()), 4), "nohist_ms_p50": round(float(np.median(allo)), 4),
            "nohist_ms_p99": round(float(np.percentile(allo, 99)), 4),
            "plan_flips": with_h[-1]["flips"], "overlapped_frames": with_h[-1]["overlapped"], "order_dropped_frames": with_h[-1]["dropped"],
            "reordered_frames": with_h[-1]["reordered"], "nohist_overlapped_frames": without[-1]["overlapped"]}


def cpu_config1():
    """BASELINE config 1: 1 000 isotropic Gaussians, 128 x 128, the C++ oracle on ONE thread."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    scene = scenes.isotropic_scene(1000, 42)
    cam = camera.default_camera(128, 128)
    cpu_oracle.forward(scene, cam, threads=1)
    best = 1e9
    for _ in range(5):
        t = {}
        cpu_oracle.forward(scene, cam, threads=1, timings=t)
        best = min(best, t["total_s"])
    return {"msplats_per_s": round(1000 / best / 1e6, 4), "ms": round(best * 1e3, 4), "threads": 1,
            "what": "BASELINE config 1: 1 000 isotropic Gaussians (seed 42), 128x128, oracle/gsr_oracle.cpp single-threaded, best of 5"}


def cpp_caller(cam, n_splats, seed, steps, warmup):
    """The same frame timed from a C++ caller: harness/gsr_harness --bench (gscuda::forward through include/gscuda_shim.hpp with
    the reference caller's device sync per frame, CudaBuffer.hpp:8-12), run as a child process after the timed region. The
    harness generates the garden-like stand-in itself (scenes.py restated in C++: R may differ in the last places)."""
    import tempfile
    exe = os.path.join(ROOT, "harness", "gsr_harness")
    if not os.path.exists(exe):
        return {"error": "harness/gsr_harness is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "camera.bin")
        with open(path, "wb") as f:
            np.array([cam.width, cam.height], np.int32).tofile(f)
            np.concatenate([cam.view, cam.proj, cam.cam_pos, [cam.tan_fovx, cam.tan_fovy], [0.0, 0.0, 0.0]]).astype(np.float32).tofile(f)
        try:
            out = subprocess.run([exe, "--bench", str(steps), path, str(n_splats), str(seed), str(warmup)], capture_output=True,
                                 text=True, timeout=300)
        except subprocess.TimeoutExpired:
            return {"error": "gsr_harness --bench timed out"}
    if out.returncode != 0:
        return {"error": (out.stderr or out.stdout).strip()[-300:]}
    fields = dict(kv.split("=") for kv in out.stdout.split() if "=" in kv)
    return {"ms_per_step": float(fields["bench_ms"]), "best_ms": float(fields["best_ms"]), "num_rendered": int(fields["numRendered"]),
            "steps": steps, "warmup": warmup,
            "what": "harness/gsr_harness --bench: a C++ caller of gscuda::forward (shim over the C ABI), device sync per frame"}


def alg_bytes(m, N, W, H, grid_x, sh_floats_read, colors_precomp=False):
    """ALGORITHMIC bytes per launch on THIS rank (SURVEY.md §8d / BASELINE.md §2). N splats, V visible, R instances,
    R_f records staged by the blend, P pixels, T tiles of this rank's band."""
    rows = m["tile_rows"]
    px_rows = min(rows[1] * 16, H) - min(rows[0] * 16, H)
    P_loc, T_loc = px_rows * W, (rows[1] - rows[0]) * grid_x
    V, R, r_f = m["visible_band"], m["num_rendered"], m["records_staged"]
    blocks = m["plan"] == "blocks"
    return {
        # (colorsPrecomp: neither the SH read nor the 12-byte rgb record of a visible Gaussian)
        "preprocess": N * (52 + (0 if colors_precomp else 4 * sh_floats_read) + 8) + V * (60 if colors_precomp else 72),
        "scan": 8 * N,
        # sort plan: the emission kernel of SURVEY.md §8d. block plan: block_emit_kernel writes the
        # SORTED pairs once (12 B each); its block-list reads (12 B per entry, E <= R) are not counted.
        "duplicate": (12 * R if m["lists_written"] else 0) if blocks else 8 * N + 20 * V + 12 * R,
        "sort_pass1": 0 if blocks else 24 * R,   # one onesweep launch: 12 B read + 12 B written per pair
        "sort_pass2": 24 * R,
        "ranges": 0 if blocks else 16 * T_loc * max(1, int(np.ceil(np.log2(max(R, 2))))) + 8 * T_loc,   # two binary searches per tile
        "blend": 40 * r_f + 20 * P_loc + 8 * T_loc,
    }


def kernel_table(m, alg):
    kernels = {}
    for k, b in alg.items():
        t = m["stage_ms"].get(k, 0.0)
        if t > 0:
            gbs = b / (t * 1e-3) / 1e9
            kernels[k] = {"ms": round(t, 4), "alg_bytes": int(b), "gbs": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
    return kernels


def load_profile_json(*names):
    for nm in names:
        path = os.path.join(ROOT, "profiles", nm)
        if os.path.exists(path):
            return json.load(open(path)), "profiles/" + nm
    return {}, None


# SIMD cycles one wave-instruction of a class holds a gfx950 SIMD for once several waves share it (scripts/micro/valu_issue.hip,
# profiles/r06_micro_valu_issue.txt: kernel wall time at 4 and 8 waves per SIMD, the launch's ends included): the THROUGHPUT
# cost an issue roofline is made of. (A wave ALONE on its SIMD issues every 5 cycles whatever the class — 8 for the
# transcendentals and the conversions, 25 to 33 for a compare whose mask a scalar instruction then reads: what a frame's last,
# lone waves run at, and what the blend's deep tiles are there to avoid. That figure is `valu_frac_one_wave`, not a roofline.)
# ("other": what no class counter claims — compares into a scalar pair and selects on one at 4.0, moves and bit operations at 2.3)
VALU_CYCLES = {"f32": 2.7, "trans_f32": 7.5, "f64": 4.3, "cvt": 4.1, "int": 2.7, "other": 3.3, "mix": 3.1}
VALU_CLASS_COUNTERS = {"f32": ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"), "trans_f32": ("SQ_INSTS_VALU_TRANS_F32",),
                       "f64": ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"), "cvt": ("SQ_INSTS_VALU_CVT",),
                       "int": ("SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64")}


def blend_issue_fractions(pmc, blend_ms):
    """Vector-issue fraction of the blend kernel from a committed PMC pass (profiles/): the wave-instructions of every class
    (SQ_INSTS_VALU_* per launch; what no class counter claims — compares into scalar masks, selects, packed operations, moves —
    is priced as "other") x the class's throughput cycles, over the 1024 SIMDs x clock x the LIVE launch time. Without the class
    counters (an older pass): SQ_INSTS_VALU x the measured mean of the blend's own mix. The counters are per launch of the same
    kernel on the same frame."""
    if not pmc or blend_ms <= 0 or "SQ_INSTS_VALU" not in pmc:
        return {}
    clock_hz = float(pmc.get("clock_ghz", 2.4)) * 1e9
    valu, salu = float(pmc["SQ_INSTS_VALU"]), float(pmc.get("SQ_INSTS_SALU", 0.0))
    avail = N_SIMD * clock_hz * blend_ms * 1e-3
    classes = {c: sum(float(pmc.get(n, 0.0)) for n in names) for c, names in VALU_CLASS_COUNTERS.items()}
    have_classes = any(n in pmc for names in VALU_CLASS_COUNTERS.values() for n in names)
    if have_classes:
        classes["other"] = max(0.0, valu - sum(classes.values()))
        cycles = sum(classes[c] * VALU_CYCLES[c] for c in classes)
        model = "sum over classes of SQ_INSTS_VALU_* x throughput cycles (VALU_CYCLES: profiles/r06_micro_valu_issue.txt)"
    else:
        cycles = valu * VALU_CYCLES["mix"]
        model = "SQ_INSTS_VALU x 3.1 cycles, the measured throughput cost of the blend's instruction mix (profiles/r06_micro_valu_issue.txt)"
    out = {"valu_frac": round(cycles / avail, 4), "valu_frac_one_wave": round(valu * 5.1 / avail, 4),
           "scalar_share": round(salu / max(valu + salu, 1.0), 4), "valu_insts": int(valu), "salu_insts": int(salu),
           "clock_ghz": float(pmc.get("clock_ghz", 2.4)),
           "valu_frac_formula": model + " / (1024 SIMDs x clock x launch time); counters from the committed PMC pass, time live; "
                                "valu_frac_one_wave: the same instructions at the 5.1 cycles a wave ALONE on its SIMD needs per instruction "
                                "(no roofline: what the kernel would take if every wave ran alone)"}
    if have_classes:
        out["valu_classes"] = {c: int(v) for c, v in classes.items()}
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
        if k in pmc and pmc[k] is not None:
            out[k.lower()] = float(pmc[k])
    if pmc.get("SQ_WAVE_CYCLES"):
        # (quad-cycles summed over the waves: waves resident per SIMD on average while the kernel runs; the kernels are built for five)
        out["mean_waves_per_simd"] = round(float(pmc["SQ_WAVE_CYCLES"]) * 4.0 / avail, 3)
    return out


def main() -> int:
    args = parse_args()
    forced = os.environ.get("GSR_FORCE_DIST") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not forced:
        return self_launch(args)
    if args.dry_run:
        return dry_run(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not forced:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or drop the launcher and let "
              f"--gpus start the ranks)", file=sys.stderr)
        return 2
    # GSR_FORCE_DIST=1 runs the sharded code path (RCCL broadcast, band exchange, re-cut) even with one
    # rank: the only way to exercise it on a single-GPU box.
    distributed = world > 1 or forced
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} needs HIP device {local_rank}; {torch.cuda.device_count()} visible "
              f"(the rasterizer has no CPU path)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)
        world = dist.get_world_size()

    from gsrast_amd import camera

    W, H = args.width, args.height
    inria = args.semantics == "inria"
    full_sh = inria and args.sh_degree > 0
    scene = None
    if rank == 0 or not distributed:
        ply_path = args.ply if (args.ply and os.path.exists(args.ply)) else ""
        if args.ply and not ply_path:
            print(f"bench.py: --ply {args.ply} does not exist: using the synthetic stand-in", file=sys.stderr)
        scene, near, far, pos, label = make_scene(args.scene, args.splats, device, full_sh, ply_path)
        if args.pose:
            pos = tuple(float(v) for v in args.pose.split(","))
        if args.opacity_scale != 1.0:
            scene["opacities"] = scene["opacities"] * float(args.opacity_scale)
            label += f", opacities x {args.opacity_scale}"
        meta = [near, far, *pos, 1.0 if label.startswith("ply:") else 0.0]
    else:
        meta, label = [0.0] * 6, ""
    if distributed:
        from gsrast_amd import sharding
        m = torch.tensor(meta, dtype=torch.float64, device=device)
        dist.broadcast(m, 0)
        near, far, pos = float(m[0]), float(m[1]), tuple(float(v) for v in m[2:5])
        from_file = bool(m[5].item() != 0.0)
        dev_scene = sharding.broadcast_scene(scene, device, 0)          # RCCL broadcast of the SoA
    else:
        from_file = label.startswith("ply:")
        dev_scene = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v)).to(device))
                     for k, v in scene.items()}
    cam = camera.default_camera(W, H, near=near, far=far, position=pos)

    run = Runner(dev_scene, W, H, device, distributed, args)
    n_splats = run.rast.num_gaussians
    if args.backward:
        assert not distributed, "--backward is a single-GPU configuration"
        run.dl_dout = torch.randn((3, H, W), generator=torch.Generator(device="cpu").manual_seed(7)).to(device)
    draw_kw = dict(plan=args.plan, overlap_emit=True if args.overlap else (False if args.serial_emit else None),
                   semantics=args.semantics, sh_degree=args.sh_degree,
                   sorted_lists=not args.no_sorted_lists, colors_precomp=args.colors_precomp, tile_history=not args.no_tile_history,
                   deep_tiles=False if args.no_deep_tiles else None)
    m = run.measure(cam, args.steps, args.warmup, **draw_kw)

    # (extras and the committed PMC figures belong to the default workload; with --ply the frame is the file's)
    default_frame = (not from_file and args.scene == "garden_like" and args.splats == DEFAULT_SPLATS and (W, H) == (1920, 1080) and not args.pose
                     and args.opacity_scale == 1.0 and not inria and not args.backward and args.plan == "auto"
                     and not args.overlap and not args.no_sorted_lists and not args.colors_precomp and not args.no_tile_history
                     and not args.no_deep_tiles)
    extras = {}
    if not args.no_extras and default_frame:
        # (6 warm-up frames: the library notices within five that a new viewpoint ends on a few slow tiles — GSR_FLAG_NO_TILE_HISTORY)
        short = dict(steps=max(5, min(args.steps, 15)), warmup=6)
        if not distributed:
            # (a) the same scene from outside the cloud (">= 1 in-scene pose" besides the default one)
            cam_out = camera.default_camera(W, H, near=near, far=far, position=(0.0, 0.0, -14.0))
            e = run.measure(cam_out, **short, **draw_kw)
            extras["pose_outside"] = brief(e, n_splats, "eye (0,0,-14): the whole cloud in front of the camera, small splats, deep lists")
            # (a') from far away: few instances per visible Gaussian (R/V = 5), the regime of the sort plan
            cam_far = camera.default_camera(W, H, near=near, far=far, position=(0.0, 0.0, -30.0))
            e = run.measure(cam_far, **short, **draw_kw)
            extras["pose_far"] = brief(e, n_splats, "eye (0,0,-30): the cloud covers a part of the screen, one or two tiles per splat (sort plan)")
            # (b) forward-only callers: GSR_FLAG_NO_SORTED_LISTS on the headline frame
            e = run.measure(cam, **short, **{**draw_kw, "sorted_lists": False})
            extras["no_sorted_lists"] = brief(e, n_splats, "headline frame with GSR_FLAG_NO_SORTED_LISTS (block plan: the 12 R bytes of sorted keys / values are not written)")
            # (b-) GSR_FLAG_NO_TILE_HISTORY on the headline frame: what a first frame, or a frame after a camera cut, takes
            e = run.measure(cam, **short, **{**draw_kw, "tile_history": False})
            extras["no_tile_history"] = brief(e, n_splats, "headline frame with GSR_FLAG_NO_TILE_HISTORY: patch order, and (without the tile times) the blend never beside the emission")
            # (b'') GSR_FLAG_SERIAL_EMIT on the headline frame: the kernels one after the other, their times those of kernels alone
            e = run.measure(cam, **short, **{**draw_kw, "overlap_emit": False})
            extras["serial_emit"] = brief(e, n_splats, "headline frame with GSR_FLAG_SERIAL_EMIT: by default the library runs the blend (vector-issue-bound) on a second "
                                                       "stream beside the emission (HBM-write-bound) where it expects the blend to be the shorter of the two; same "
                                                       "outputs. Each kernel runs longer while they share the chip: these are the kernels ALONE")
            # (b') the caller-side route the reference's signature offers around the 192-byte-stride DC read: colorsPrecomp
            e = run.measure(cam, **short, **{**draw_kw, "colors_precomp": True})
            b = brief(e, n_splats, "headline frame with the colours passed as colorsPrecomp (GSCuda.cuh:111; computed once per scene, "
                                   "0.5 + 0.4 DC is view-independent): same pixels bit for bit, the preprocess reads no SH and writes no rgb")
            pb = alg_bytes(e, n_splats, W, H, run.grid_x, 3, colors_precomp=True)["preprocess"]
            pms = e["stage_ms"].get("preprocess", 0.0)
            b["preprocess"] = {"ms": round(pms, 4), "alg_bytes": int(pb), "gbs": round(pb / max(pms, 1e-9) / 1e6, 1),
                               "frac_hbm": round(pb / max(pms, 1e-9) / 1e6 / HBM_PEAK_GBS, 4)}
            extras["colors_precomp"] = b
            # (c) blend-bound variant: opacities x 0.1, so pixels saturate late and the blend stages most of R
            saved = run.rast.opacities
            run.rast.opacities = (saved * 0.1).contiguous()
            e = run.measure(cam, **short, **draw_kw)
            b = brief(e, n_splats, "headline frame with opacities x 0.1 (R_f ~ R: the blend walks deep lists)")
            b["blend_ms_per_million_staged_records"] = round(e["stage_ms"].get("blend", 0.0) / max(e["records_staged"], 1) * 1e6, 5)
            extras["blend_bound"] = b
            e = run.measure(cam_out, **short, **draw_kw)
            b = brief(e, n_splats, "eye (0,0,-14) with opacities x 0.1")
            b["blend_ms_per_million_staged_records"] = round(e["stage_ms"].get("blend", 0.0) / max(e["records_staged"], 1) * 1e6, 5)
            extras["blend_bound_pose_outside"] = b
            run.rast.opacities = saved
            # (d) BASELINE config 5: forward + backward on the headline frame, with the sorted lists and without them
            # (gsr_backward then walks the block lists the forward left)
            run.dl_dout = torch.randn((3, H, W), generator=torch.Generator(device="cpu").manual_seed(7)).to(device)
            e = run.measure(cam, **short, **draw_kw)
            extras["forward_backward"] = brief(e, n_splats, "BASELINE config 5: forward + backward (all gradients down to the inputs) on the headline frame")
            # (every output of gsr_backward is optional: the chain is bound by what it writes, 150-odd bytes per Gaussian in all)
            run.bw_outputs = ("dL_dmean2D", "dL_dconic_opacity", "dL_dshs", "dL_dmeans3D", "dL_dscales", "dL_drotations")
            e = run.measure(cam, **short, **draw_kw)
            extras["forward_backward_inputs"] = brief(e, n_splats, "the same, only the gradients of the inputs asked for (means3D, scales, rotations, opacity = dL_dconic_opacity.w, SH) and dL_dmean2D; dL_dcolors / dL_dcov2D / dL_dcov3D not written")
            run.bw_outputs = ("dL_dmean2D", "dL_dcov3D", "dL_dshs")
            e = run.measure(cam, **short, **draw_kw)
            extras["forward_backward_config5"] = brief(e, n_splats, "the same, exactly the outputs BASELINE config 5 names: dL_dmean2D, dL_dcov3D, dL_dshs")
            run.bw_outputs = None
            e = run.measure(cam, **short, **{**draw_kw, "sorted_lists": False})
            extras["forward_backward_no_sorted_lists"] = brief(e, n_splats, "the same with GSR_FLAG_NO_SORTED_LISTS: the backward reads the tile lists from the block lists")
            run.dl_dout = None
            # (e) what a viewer and a trainer get: a camera path and a sequence of unrelated views, every frame rendered once
            extras["path"] = time_sequence(run, walk_path(W, H, near, far), draw_kw)
            extras["path"]["what"] = ("120 poses of the reference's first-person camera walking from (0,0,-30) to the default pose and turning there, each "
                                      "rendered once, device sync per frame; with this rasterizer's tile history / with GSR_FLAG_NO_TILE_HISTORY")
            extras["random_views"] = time_sequence(run, random_views(W, H, near, far), draw_kw, backward=True)
            extras["random_views"]["what"] = ("60 unrelated cameras of one size, forward + backward (every array) each, rendered once in turn: BASELINE "
                                              "config 5's access pattern; with the tile history (which drops its order) / without")
        else:
            # BASELINE config 3: the same scene at 3840 x 2160, tile rows sharded over the ranks
            run4k = Runner(dev_scene, 3840, 2160, device, distributed, args)
            cam4k = camera.default_camera(3840, 2160, near=near, far=far, position=pos)
            e = run4k.measure(cam4k, **short, **draw_kw)
            c3 = brief(e, n_splats, "BASELINE config 3: same scene and pose at 3840x2160, tile rows sharded over the ranks")
            c3["per_rank"], c3["bands"] = e["per_rank"], e["bands"]
            extras["config3_4k"] = c3
            run4k.exch.close()
            del run4k

    if rank == 0:
        grid_x = run.grid_x
        ms_per_step = m["ms_per_step"]
        stage_ms = dict(m["stage_ms"])
        blocks = m["plan"] == "blocks"
        # (colours written beside the depth sort: the preprocess kernel's own bytes are those of the colorsPrecomp route)
        alg = alg_bytes(m, n_splats, W, H, grid_x, 48 if full_sh else 3, colors_precomp=args.colors_precomp or m.get("colors_beside", False))
        kernels = kernel_table(m, alg)
        dom = max(("sort_pass1", "sort_pass2", "blend", "duplicate", "preprocess", "ranges"), key=lambda k: stage_ms.get(k, 0.0))
        dom_names = {"sort_pass1": "onesweep_kernel<u64> (tile-column digit pass)",
                     "sort_pass2": "onesweep_kernel<u64> (tile-row digit pass)",
                     "blend": ("blend_blocks_kernel" if blocks and not m["blend_from_lists"] else
                               ("blend_deep_kernel (four waves per tile)" if m.get("deep_tiles") else "blend_wave_kernel")),
                     "duplicate": "block_emit_kernel (sorted lists written directly)" if blocks else "emit_chunk_kernel",
                     "preprocess": "preprocess_inria_kernel" if inria else "preprocess_kernel", "ranges": "tile_ranges_search_kernel"}

        # HBM bytes and instruction counts per launch measured with PMC counters in separate rocprofv3 passes of this
        # same command (profiles/): valid only for the default single-GPU workload they were taken on.
        traffic, traffic_src, blend_pmc, tj = {}, None, {}, {}
        if default_frame and not distributed:
            tj, traffic_src = load_profile_json("pmc_traffic_r06.json", "pmc_traffic_r05.json", "pmc_traffic_r04.json", "pmc_traffic_r03.json", "pmc_traffic_r02.json", "pmc_traffic_r01.json")
            traffic = tj.get("blocks" if blocks else "sort", {})
            blend_pmc = tj.get("blend_insts", {})

        def roof(k, note):
            e = kernels.get(k, {"gbs": 0.0, "ms": 0.0, "alg_bytes": 0})
            return {"bound": "hbm", "kernel": dom_names[k], "achieved": e["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(e["gbs"] / HBM_PEAK_GBS, 5), "traffic": traffic.get(k),
                    "traffic_source": f"{traffic_src} (rocprofv3 --pmc, separate passes)" if k in traffic else None,
                    "algorithmic_bytes_per_launch": e["alg_bytes"], "avg_launch_ms": e["ms"], "note": note}

        if args.backward:
            stage_ms["render_backward"], stage_ms["chain_backward"] = m["backward_ms"]
        rb = roof("blend", "the kernel BASELINE.json names; bound by vector-instruction issue, not HBM (~100 flop/B): the HBM "
                           "fraction is reported as measured, the issue fraction beside it"
                           + ("; it ran beside the emission here (both take longer while they share the chip): alone, roofline_blend_alone"
                              if m.get("emit_overlapped") else ""))
        rb.update(blend_issue_fractions(blend_pmc, stage_ms.get("blend", 0.0)))
        rb["ms_per_million_staged_records"] = round(stage_ms.get("blend", 0.0) / max(m["records_staged"], 1) * 1e6, 5)
        out = {
            "metric": "forward_backward_msplats_per_s" if args.backward else "forward_msplats_per_s",
            "value": round(n_splats / (ms_per_step * 1e-3) / 1e6, 3),
            "unit": "Msplats/s",
            "fps": round(1e3 / ms_per_step, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_effective": m["untimed_frames"],      # spin-up + the counted frame + the W warm-up steps: every frame rendered before the timed region
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median": round(m["ms_median"], 4), "ms_per_step_p99": round(m["ms_p99"], 4),   # (BASELINE.md section 2 defines the median)
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": label.split(" ")[0] if label.startswith("ply:") else "synthetic",
            # (the driver's record keeps the first 24 keys of `config`, flat scalars only: the measured figures come first —
            # filled in below, in this order — and what describes the run without measuring anything is under "detail")
            "config": {"workload": f"{label}, {W}x{H} forward, {args.semantics} semantics"
                                   + (f" (SH degree {args.sh_degree})" if inria else " (DC colour)")
                                   + (", colours passed as colorsPrecomp" if args.colors_precomp else "")
                                   + f", camera at {tuple(round(v, 3) for v in pos)}"
                                   + (" (the reference's default pose)" if not args.pose and args.scene != "stress" else ""),
                       "splats": n_splats, "num_rendered": m["num_rendered_total"], "records_staged": m["records_staged_total"],
                       "binning_plan": m["plan"], "ms_median": round(m["ms_median"], 4), "ms_p99": round(m["ms_p99"], 4),
                       "untimed_frames": m["untimed_frames"]},
            "detail": {"width": W, "height": H, "visible": m["visible"],
                       "minstances_per_s": round(m["num_rendered_total"] / (ms_per_step * 1e-3) / 1e6, 2),
                       "sorted_lists_written": m["lists_written"], "blend_from_sorted_lists": m["blend_from_lists"],
                       "slow_tiles_first": m.get("tiles_reordered", False), "emit_overlapped": m.get("emit_overlapped", False),
                       "colors_beside_depth_sort": m.get("colors_beside", False), "spinup_frames": max(0, int(args.spinup)),
                       "deep_tiles": m.get("deep_tiles", False),
                       "band_exchange": ({"transport": run.exch.transport, "gather": args.gather, "note": run.exch.transport_note}
                                         if distributed else None),
                       "bands": m["bands"], "per_rank": m["per_rank"]},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_meaning": ({"depth_order": "visible-key compaction + depth sort + block lists"
                                                 + (" (geomState.rgb is written beside them on the second stream: +0.05 ms here for -0.10 ms in preprocess; "
                                                    "alone: the serial_emit entry)" if m.get("colors_beside") and n_splats <= (1 << 24) else ""),
                                  "sort_pass1": "unit masks + prefixes + tile ranges",
                                  "duplicate": "block_emit_kernel (the sorted keys / values)",
                                  "blend": (("blend_deep_kernel" if m.get("deep_tiles") else "blend_wave_kernel") + " (from the sorted lists)") if m["blend_from_lists"] else "blend_blocks_kernel"} if blocks else
                                 {"depth_order": "visible-key compaction + depth sort + column counts / scan", "duplicate": "emit_chunk_kernel",
                                  "sort_pass2": "onesweep pass on the tile row", "blend": "blend_deep_kernel" if m.get("deep_tiles") else "blend_wave_kernel"}),
            "stage_ms_source": (f"HIP events recorded by the library on the launching stream (GSR_FLAG_PROFILE) over {m['prof_steps']} "
                                f"frames of the same workload run right after the timed region; those frames took "
                                f"{m['profiled_ms_per_step']:.4f} ms each with the events in place"),
            "roofline": roof(dom, "dominant kernel of this frame; HIP-event time of the launch on its own stream"
                                  + ("; the blend ran beside it on the library's second stream, as in the timed frames (the kernel "
                                     "alone: roofline_alone, GSR_FLAG_SERIAL_EMIT)" if m.get("emit_overlapped") and dom == "duplicate" else "")),
            "roofline_blend": rb,
            "kernels": kernels,
        }
        # what the driver keeps of this line is its head: the representative frames and the whole-frame rate go into `config`
        # (flat scalars only; at most 24 keys: tests/test_gpu_bench.py)
        cfg, detail = out["config"], out["detail"]
        for key, name in (("no_sorted_lists", "ms_no_sorted_lists"), ("serial_emit", "ms_serial"), ("no_tile_history", "ms_no_tile_history"),
                          ("blend_bound", "ms_faint"), ("pose_outside", "ms_pose_outside"), ("pose_far", "ms_pose_far"),
                          ("forward_backward", "ms_forward_backward"), ("config3_4k", "ms_config3_4k")):
            if key in extras:
                cfg[name] = extras[key]["ms_per_step"]
        if "colors_precomp" in extras:
            detail["ms_colors_precomp"] = extras["colors_precomp"]["ms_per_step"]
        if "path" in extras:
            pth, rv = extras["path"], extras["random_views"]
            cfg.update({"path_ms_p50": pth["ms_p50"], "path_ms_p99": pth["ms_p99"], "path_ms_nohist_p50": pth["nohist_ms_p50"],
                        "randview_ms": rv["ms_mean"], "randview_ms_nohist": rv["nohist_ms_mean"]})
            detail.update({"path_ms_mean": pth["ms_mean"], "path_ms_nohist_p99": pth["nohist_ms_p99"],
                           "path_hist_over_nohist": pth["hist_over_nohist"], "randview_hist_over_nohist": rv["hist_over_nohist"],
                           "path_plan_flips": pth["plan_flips"], "path_overlapped_frames": pth["overlapped_frames"],
                           "path_order_dropped_frames": pth["order_dropped_frames"], "randview_ms_p50": rv["ms_p50"],
                           "randview_ms_nohist_p50": rv["nohist_ms_p50"], "randview_order_dropped_frames": rv["order_dropped_frames"]})
        cfg["frame_alg_gbs"] = round(sum(v["alg_bytes"] for v in kernels.values()) / (ms_per_step * 1e-3) / 1e9, 1)   # all stages' algorithmic bytes / frame time
        if "valu_frac" in rb:
            cfg["blend_valu_frac"] = rb["valu_frac"]
        cfg["parallelism"] = f"tile-rows x{world}" if distributed else "single GPU"
        cfg["rccl_ranks"] = world if distributed else 0
        if distributed:
            # (sharded runs measure few of the frames above: room for what tells an imbalance from a slow transport, as strings)
            cfg["bands"] = ",".join(f"{a}-{b}" for a, b in zip(m["bands"][:-1], m["bands"][1:])) if m["bands"] else ""
            cfg["per_rank_render_ms"] = ",".join(str(r["render_ms"]) for r in (m["per_rank"] or []))
            cfg["per_rank_exchange_ms"] = ",".join(str(r["exchange_ms"]) for r in (m["per_rank"] or []))
            cfg["band_exchange"] = run.exch.transport
        if distributed:
            proj, src = load_profile_json("band_projection.json")
            key = f"{W}x{H}"
            if default_frame and key in proj and str(world) in proj[key].get("recut_ms", {}):
                cfg["projected_ms"] = proj[key]["recut_ms"][str(world)]       # (the slowest band of the re-cut frame on one GPU; exchange not included)
                detail["projected_ms"] = {"slowest_band_ms": proj[key]["recut_ms"][str(world)], "one_gpu_ms": proj[key]["recut_ms"].get("1"),
                                          "source": f"{src}: one GPU rendering each band of the re-cut frame in turn; exchange not included"}
        elif default_frame and not args.no_extras:
            cpp = cpp_caller(cam, n_splats, 43, max(5, min(args.steps, 30)), 5)
            out["cpp_caller_ms"] = cpp.get("ms_per_step")
            out["cpp_caller"] = cpp
        out.update(extras)
        # the blend's issue fraction on the frames it is the long pole of (counters of the same frames' committed PMC passes)
        for key, pm in (("pose_far", "blend_insts_far"), ("pose_outside", "blend_insts_outside"), ("blend_bound", "blend_insts_bound")):
            if key in extras and default_frame and not distributed and pm in tj:
                fr = blend_issue_fractions(tj[pm], extras[key]["stage_ms"].get("blend", 0.0))
                extras[key]["blend_issue"] = {k: fr[k] for k in ("valu_frac", "valu_frac_one_wave", "mean_waves_per_simd", "valu_insts") if k in fr}
                extras[key]["blend_issue"]["kernel"] = tj[pm].get("kernel")
        if "serial_emit" in extras and dom == "duplicate":
            # the dominant kernel with nothing beside it (the frames of GSR_FLAG_SERIAL_EMIT above)
            alone_ms = extras["serial_emit"]["stage_ms"].get("duplicate", 0.0)
            if alone_ms > 0:
                gbs = kernels["duplicate"]["alg_bytes"] / (alone_ms * 1e-3) / 1e9
                out["roofline_alone"] = {"kernel": dom_names["duplicate"], "avg_launch_ms": alone_ms, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "frame_ms": extras["serial_emit"]["ms_per_step"]}
            blend_alone_ms = extras["serial_emit"]["stage_ms"].get("blend", 0.0)
            if blend_alone_ms > 0:
                gbs = kernels["blend"]["alg_bytes"] / (blend_alone_ms * 1e-3) / 1e9
                out["roofline_blend_alone"] = {"kernel": dom_names["blend"], "avg_launch_ms": blend_alone_ms, "achieved": round(gbs, 1),
                                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                                               **blend_issue_fractions(blend_pmc, blend_alone_ms)}
        if not distributed and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, cam, args.cpu_sample)
            out["cpu_baseline"]["config1_single_thread"] = cpu_config1()
    if distributed:
        if run.exch is not None:
            run.exch.close()                  # (the library's own RCCL communicator, if the exchange ran on it)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)          # the ONE JSON line, last thing on stdout
    return 0


def brief(e, n_splats, what):
    """Compact record of an extra (untimed-for-the-headline) measurement."""
    if sum(e.get("backward_ms") or [0.0]) > 0:
        e["stage_ms"] = {**e["stage_ms"], "render_backward": e["backward_ms"][0], "chain_backward": e["backward_ms"][1]}
    return {"what": what, "ms_per_step": round(e["ms_per_step"], 4), "fps": round(1e3 / e["ms_per_step"], 2),
            "msplats_per_s": round(n_splats / (e["ms_per_step"] * 1e-3) / 1e6, 3), "num_rendered": e["num_rendered_total"],
            "records_staged": e["records_staged_total"], "visible": e["visible"], "binning_plan": e["plan"],
            "blend_from_sorted_lists": e["blend_from_lists"], "slow_tiles_first": e.get("tiles_reordered", False),
            "deep_tiles": e.get("deep_tiles", False),
            "emit_overlapped": e.get("emit_overlapped", False), "colors_beside_depth_sort": e.get("colors_beside", False),
            "sorted_lists_written": e["lists_written"], "stage_ms": {k: round(v, 4) for k, v in e["stage_ms"].items() if v > 0}}


if __name__ == "__main__":
    sys.exit(main())
