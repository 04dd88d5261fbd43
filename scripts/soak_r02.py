"""Round-2 soak: random first-person poses. Every frame under the block plan, the block plan with
GSR_FLAG_NO_SORTED_LISTS and the sort plan: pixels, finalT, nContrib, ranges and the staged-record count must agree bit
for bit (and the sorted keys / values of the two plans that write them). Every `bw_every`-th pose also runs gsr_backward
from the three forward states (sorted lists of the sort plan; block plan: block lists for the shallow blocks + sorted
lists; block plan without sorted lists: block lists only): the sums the render backward leaves (dL_dmean2D, dL_dconic_opacity, dL_dcolors) must agree to within a
small multiple of what two runs of ONE path differ by (float atomics land in a different order every run); the chain
outputs computed from them are reported.
Usage: python scripts/soak_r02.py [poses] [bw_every] [width height] [gscuda|inria]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gsrast_amd import camera, scenes
from gsrast_amd.rasterizer import SplatRasterizer

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bw_every = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
SEM = sys.argv[5] if len(sys.argv) > 5 else "gscuda"
scene = scenes.garden_like_scene(1_500_000, seed=91)
span = float(np.max(scene["means3D"][:, :3].max(0) - scene["means3D"][:, :3].min(0)))
r = SplatRasterizer(W, H, background=(0.1, 0.2, 0.3))
r.configure_from_scene(scene)
rng = np.random.default_rng(6)
dl = torch.randn((3, H, W), generator=torch.Generator().manual_seed(3)).cuda()


def lists_digest():
    b = r.map_binning_state()
    k, v = b["keys"], b["values"].to(torch.int64)
    w = torch.arange(k.numel(), device=k.device, dtype=torch.int64) * 0x9E3779B1 + 12345
    return int((k ^ w).sum()), int((v * (w | 1)).sum())


def frame_state(img):
    st = r.map_image_state()
    return img.clone(), st["ranges"].clone(), st["nContrib"].clone(), st["finalT"].clone(), r.last_records_staged, r.last_num_rendered


def same(a, b):
    return all(torch.equal(x, y) if isinstance(x, torch.Tensor) else x == y for x, y in zip(a, b))


def grads():
    return {k: v.clone() for k, v in r.backward(dl, semantics=SEM).items() if isinstance(v, torch.Tensor)}


bad = frames = bw_frames = 0
worst = 0.0
BASE = ("dL_dmean2D", "dL_dconic_opacity", "dL_dcolors")
chain_worst, chain_noise = {}, {}
for i in range(poses):
    pos = tuple(float(x) for x in rng.uniform(-6, 6, 3))
    cam = camera.first_person_camera(pos, float(rng.uniform(-3.14, 3.14)), float(rng.uniform(-0.8, 0.8)), float(np.radians(45.0)),
                                     0.001 * span, span, W, H, True)
    do_bw = i % bw_every == 0
    s_sort = frame_state(r.draw(cam, plan="sort", count_staged=True, semantics=SEM))
    if s_sort[5] == 0:
        continue
    frames += 1
    d_sort = lists_digest()
    g_sort = grads() if do_bw else None
    g_sort2 = grads() if do_bw else None
    s_blk = frame_state(r.draw(cam, plan="blocks", count_staged=True, semantics=SEM))
    d_blk = lists_digest()
    g_blk = grads() if do_bw else None
    s_nol = frame_state(r.draw(cam, plan="blocks", count_staged=True, sorted_lists=False, semantics=SEM))
    assert not r.last_lists_written
    g_nol = grads() if do_bw else None
    ok = same(s_sort, s_blk) and same(s_sort, s_nol) and d_sort == d_blk
    note = ""
    if do_bw:
        bw_frames += 1
        for k, v in g_sort.items():
            noise = float((g_sort2[k] - v).abs().max())
            scale = float(v.abs().max()) + 1e-30
            for name, g in (("blocks", g_blk), ("blocks, no sorted lists", g_nol)):
                err = float((g[k] - v).abs().max())
                if k in BASE:
                    # the sums the render backward leaves: every path adds the same terms, in its own order
                    worst = max(worst, err / scale)
                    if err > max(4.0 * noise, 2e-4 * scale):
                        ok = False
                        note += f" [{k} / {name}: err {err:.3g} noise {noise:.3g} scale {scale:.3g}]"
                else:
                    # chain outputs: a per-Gaussian function of those sums, ill-conditioned for Gaussians that fill the
                    # screen a few centimetres from the camera (two runs of ONE path differ by 10 % there): reported only
                    chain_worst[k] = max(chain_worst.get(k, 0.0), err / scale)
                    chain_noise[k] = max(chain_noise.get(k, 0.0), noise / scale)
    bad += 0 if ok else 1
    if i % 20 == 0 or not ok:
        print(f"pose {i}: R={s_sort[5]} R_f={s_sort[4]} ok={ok} lists {d_sort}{' backward checked' if do_bw else ''}{note}", flush=True)
print(f"{frames} frames with R > 0, {bw_frames} with backward, {bad} mismatches; worst difference between paths in the render "
      f"backward's sums: {worst:.2e} of the largest")
for k in chain_worst:
    print(f"  chain output {k}: worst difference between paths {chain_worst[k]:.2e} of the largest; between two runs of the "
          f"sorted-list path {chain_noise[k]:.2e}")
sys.exit(1 if bad else 0)
