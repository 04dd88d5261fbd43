"""GPU: BASELINE.json configs 3, 4 and 5 at their workloads (the path of reference GSCuda.cu:695-811).

  config 3  5 834 784 splats (garden-like stand-in, seed 43) at 3840 x 2160, tile rows in 8 uneven bands:
            the banded frame equals the whole frame bit for bit; every 8th splat at 4K against the CPU oracle.
  config 4  50 M anisotropic splats (seed 44) at 1920 x 1080: both binning plans bit-equal, list properties,
            every 16th splat against the CPU oracle (per-Gaussian outputs of the full run too).
  config 5  forward + backward on the 5.83 M-splat scene at 1920 x 1080: finite everywhere, zero for culled
            splats, Gaussians picked at random checked against oracle/backward_np.py on the tiles they touch.

The scenes are generated on the device (gsrast_amd/scenes.py: same splitmix64 definition); the oracle's inputs
are copied FROM the device scene so both sides see the same bits.
"""
import gc

import numpy as np
import pytest

from helpers import check_backward_chain, image_report

pytestmark = pytest.mark.gpu


def _release():
    import torch
    gc.collect()
    torch.cuda.empty_cache()


def _threads():
    from oracle import cpu_oracle
    return max(1, min(64, cpu_oracle.hardware_concurrency()))


def _compare_with_oracle(r, exp, img, what):
    """Lists bit for bit, pixels within 1e-4 (threshold flips bounded), per-Gaussian floats bit for bit."""
    assert r.last_num_rendered == exp["num_rendered"], what
    assert r.last_records_staged == exp["records_staged"], what
    g = r.map_geometry_state()
    vis = exp["tilesTouched"] > 0               # (the per-Gaussian arrays are only written for visible Gaussians)
    for k in ("means2D", "depths", "cov3D", "rgb", "conicOpacity"):
        assert np.array_equal(g[k].cpu().numpy()[vis], exp[k][vis]), (what, k)
    assert np.array_equal(g["radii"].cpu().numpy(), exp["radii"]), what
    assert np.array_equal(g["tilesTouched"].cpu().numpy().view(np.uint32), exp["tilesTouched"]), what
    b = r.map_binning_state()
    assert np.array_equal(b["keys"].cpu().numpy().view(np.uint64), exp["keys"]), what
    assert np.array_equal(b["values"].cpu().numpy().view(np.uint32), exp["values"]), what
    assert np.array_equal(r.map_image_state()["ranges"].cpu().numpy().view(np.uint32), exp["ranges"]), what
    max_err, n_bad, per_pixel = image_report(img, exp["out_color"], 1e-4)
    assert n_bad <= 20 and max_err <= 8e-3, (what, max_err, n_bad)        # 1e-4 abs per channel but for threshold flips
    assert np.percentile(per_pixel, 99.99) <= 1e-5, what
    nc = r.map_image_state()["nContrib"].cpu().numpy().view(np.uint32)
    assert (nc != exp["nContrib"]).sum() <= 40, what
    return max_err, n_bad


def test_config3_garden_4k_in_eight_bands_and_against_the_oracle():
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    from oracle import cpu_oracle
    W, H = 3840, 2160
    scene = scenes.garden_like_scene_device(5_834_784, seed=43, device="cuda:0")
    pos = scene["means3D"][:, :3]
    span = float((pos.max(0).values - pos.min(0).values).max())
    cam = camera.default_camera(W, H, near=0.001 * span, far=span)
    r = SplatRasterizer(W, H, background=(0.0, 0.0, 0.0))
    r.configure_from_scene(scene)
    img = r.draw(cam).clone()
    R = r.last_num_rendered
    assert R > 1_000_000_000 and r.last_plan == "blocks"          # 1.2 G instances: 28 % of the 32-bit offset range
    b = r.map_binning_state()
    assert bool((b["keys"][1:] >= b["keys"][:-1]).all()), "sorted keys out of order"
    rg = r.map_image_state()["ranges"].to(torch.int64)
    assert int((rg[:, 1] - rg[:, 0]).sum()) == R
    # tile rows in 8 uneven bands, as 8 ranks would hold them (135 tile rows)
    bands = ((0, 11), (11, 30), (30, 47), (47, 64), (64, 70), (70, 95), (95, 118), (118, 135))
    r.out_color.fill_(-1.0)
    total = 0
    for rows in bands:
        r.draw(cam, tile_rows=rows)
        total += r.last_num_rendered
    assert total == R, "bands must partition the instances"
    assert torch.equal(r.out_color, img), "banded 4K frame differs from the whole frame"
    r = b = rg = img = None
    _release()

    # every 8th splat at the same camera and resolution against the CPU oracle (R of the sample ~ 150 M)
    sub = scenes.scene_rows(scene, slice(None, None, 8))
    del scene
    exp = cpu_oracle.forward(sub, cam, (0.0, 0.0, 0.0), threads=_threads())
    r2 = SplatRasterizer(W, H, background=(0.0, 0.0, 0.0))
    r2.configure_from_scene(sub)
    img2 = r2.draw(cam, count_staged=True).cpu().numpy()
    assert exp["num_rendered"] > 100_000_000
    _compare_with_oracle(r2, exp, img2, "config 3, every 8th splat at 4K")


def test_config4_50m_stress_plans_agree_and_sample_against_the_oracle():
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    from oracle import cpu_oracle
    W, H, N = 1920, 1080, 50_000_000
    scene = scenes.stress_scene_device(N, seed=44, device="cuda:0")          # gscuda semantics read the DC triple only
    cam = camera.default_camera(W, H, near=0.1, far=100.0, position=(0.0, 0.0, -25.0))
    r = SplatRasterizer(W, H, background=(0.0, 0.0, 0.0))
    r.configure_from_scene(scene)
    img_auto = r.draw(cam, count_staged=True).clone()
    staged = r.last_records_staged
    assert r.last_plan == "sort"                                              # R / N = 2.5: below the block plan's threshold
    R = r.last_num_rendered
    assert R > 100_000_000
    b = r.map_binning_state()
    keys_s, vals_s = b["keys"].clone(), b["values"].clone()
    ranges_s = r.map_image_state()["ranges"].clone()
    img_b = r.draw(cam, plan="blocks", count_staged=True)
    assert r.last_plan == "blocks" and r.last_num_rendered == R and r.last_records_staged == staged
    b = r.map_binning_state()
    assert torch.equal(b["keys"], keys_s) and torch.equal(b["values"], vals_s), "the two binning plans differ"
    assert torch.equal(r.map_image_state()["ranges"], ranges_s)
    assert torch.equal(img_b, img_auto)
    # list properties (no oracle needed): sorted, ties by index, depth half = depth of the Gaussian, ranges partition
    g = r.map_geometry_state()
    tt = g["tilesTouched"].to(torch.int64) & 0xFFFFFFFF
    assert int(tt.sum()) == R == (int(g["pointOffsets"][-1]) & 0xFFFFFFFF)
    assert bool((keys_s[1:] >= keys_s[:-1]).all())
    tie = keys_s[1:] == keys_s[:-1]
    assert bool((vals_s[1:][tie] > vals_s[:-1][tie]).all())
    depth_bits = g["depths"].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    samp = torch.randint(0, R, (2_000_000,), device=keys_s.device, generator=torch.Generator(device=keys_s.device).manual_seed(1))
    assert bool(((keys_s[samp] & 0xFFFFFFFF) == depth_bits[vals_s[samp].to(torch.int64)]).all())
    counts = torch.bincount(keys_s >> 32, minlength=120 * 68)
    rg = ranges_s.to(torch.int64)
    assert bool(((rg[:, 1] - rg[:, 0]) == counts).all())
    del keys_s, vals_s, tie, samp, counts, depth_bits

    # every 16th splat: the full run's per-Gaussian outputs at those rows, then the sample rendered on its own
    rows = slice(None, None, 16)
    sub = scenes.scene_rows(scene, rows)
    exp = cpu_oracle.forward(sub, cam, (0.0, 0.0, 0.0), threads=_threads())
    vis = exp["tilesTouched"] > 0
    assert vis.sum() > 1_000_000
    for k in ("means2D", "depths", "cov3D", "rgb", "conicOpacity"):
        assert np.array_equal(g[k][rows].cpu().numpy()[vis], exp[k][vis]), k
    assert np.array_equal(g["radii"][rows].cpu().numpy(), exp["radii"])
    assert np.array_equal(g["tilesTouched"][rows].cpu().numpy().view(np.uint32), exp["tilesTouched"])
    del g, b, rg, img_auto, img_b
    r = None
    del scene
    _release()
    for plan in ("sort", "blocks"):
        r2 = SplatRasterizer(W, H, background=(0.0, 0.0, 0.0))
        r2.configure_from_scene(sub)
        img2 = r2.draw(cam, count_staged=True, plan=plan).cpu().numpy()
        assert r2.last_plan == plan
        _compare_with_oracle(r2, exp, img2, f"config 4, every 16th splat, plan {plan}")
        r2 = None
        _release()


def _tile_rects(means2D, ext, gx, gy):
    """getRect (GSCuda.cu:248-259) with the stored extents, in torch float32: [x0, x1) x [y0, y1) in tiles."""
    import torch
    p = means2D
    e = ext.to(torch.float32)
    x0 = ((p[:, 0] - e[:, 0]) / 16.0).to(torch.int32).clamp(0, gx)
    y0 = ((p[:, 1] - e[:, 1]) / 16.0).to(torch.int32).clamp(0, gy)
    x1 = ((p[:, 0] + e[:, 0] + 15.0) / 16.0).to(torch.int32).clamp(0, gx)
    y1 = ((p[:, 1] + e[:, 1] + 15.0) / 16.0).to(torch.int32).clamp(0, gy)
    return x0, y0, x1, y1


def test_config5_garden_1080p_forward_backward_fullsize():
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    from oracle import backward_np as B
    W, H, N = 1920, 1080, 5_834_784
    gx, gy = 120, 68
    scene = scenes.garden_like_scene_device(N, seed=43, device="cuda:0")
    pos = scene["means3D"][:, :3]
    span = float((pos.max(0).values - pos.min(0).values).max())
    cam = camera.default_camera(W, H, near=0.001 * span, far=span)
    bg = (0.1, 0.3, 0.2)
    r = SplatRasterizer(W, H, background=bg)
    r.configure_from_scene(scene)
    r.draw(cam)
    assert r.last_num_rendered > 200_000_000
    dL = torch.randn((3, H, W), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(7))
    got_dev = r.backward(dL)
    geo = r.map_geometry_state()
    culled = geo["radii"] <= 0
    for k, v in got_dev.items():
        assert bool(torch.isfinite(v).all()), k
        assert bool((v[culled] == 0).all()), (k, "culled splats must get zero gradient")
    assert int((got_dev["dL_dcolors"].abs().sum(1) > 0).sum()) > 100_000       # a real share of the scene gets gradient

    # Gaussians whose tiles all lie inside one of 40 random 2 x 2-tile windows: their gradients come from those
    # tiles only, so the float64 oracle evaluated on the windows gives their complete gradients.
    x0, y0, x1, y1 = _tile_rects(geo["means2D"], r.rects, gx, gy)
    tt = geo["tilesTouched"].to(torch.int64)
    area = ((x1 - x0) * (y1 - y0)).to(torch.int64)
    vis = tt > 0
    assert bool((area[vis] == tt[vis]).all())
    ranges = r.map_image_state()["ranges"].cpu().numpy().view(np.uint32).astype(np.int64)
    ncontrib = r.map_image_state()["nContrib"]
    # (windows whose deepest pixel walks more than 6 000 records are passed over: the oracle holds [records x 256]
    # float64 arrays per tile)
    nc_host = ncontrib.cpu().numpy().view(np.uint32)
    rng = np.random.default_rng(11)
    wins = []
    for _ in range(400):
        wx, wy = int(rng.integers(0, gx - 1)), int(rng.integers(0, gy - 1))
        if (wx, wy) not in wins and int(nc_host[wy * 16:wy * 16 + 32, wx * 16:wx * 16 + 32].max()) <= 6000:
            wins.append((wx, wy))
        if len(wins) == 40:
            break
    assert len(wins) >= 20, len(wins)
    plist = r.map_binning_state()["values"]
    acc = {"dL_dmean2D": {}, "dL_dconic": {}, "dL_dopacity": {}, "dL_dcolors": {}}
    cand_all = []
    nc_mismatch = 0
    for wx, wy in wins:
        inside = vis & (x0 >= wx) & (x1 <= wx + 2) & (y0 >= wy) & (y1 <= wy + 2)
        cand = torch.nonzero(inside).flatten().cpu().numpy()
        cand_all.append(cand)
        sums = {k: np.zeros((cand.size, d)) for k, d in (("dL_dmean2D", 2), ("dL_dconic", 3), ("dL_dopacity", 1), ("dL_dcolors", 3))}
        where = {int(i): j for j, i in enumerate(cand)}
        for ty in (wy, wy + 1):
            for tx in (wx, wx + 1):
                t = ty * gx + tx
                ya, yb, xa, xb = ty * 16, min(H, ty * 16 + 16), tx * 16, min(W, tx * 16 + 16)
                nc_tile = ncontrib[ya:yb, xa:xb]
                depth = int(nc_tile.max())                               # the list prefix that reaches every last contributor
                a = int(ranges[t, 0])
                assert depth <= int(ranges[t, 1]) - a
                ids = plist[a:a + depth].to(torch.int64)
                tile_g = np.zeros((3, 16, 16))
                tile_g[:, : yb - ya, : xb - xa] = dL[:, ya:yb, xa:xb].cpu().numpy()
                res = B.blend_tile_backward(geo["means2D"][ids].cpu().numpy(), geo["conicOpacity"][ids].cpu().numpy(),
                                            geo["rgb"][ids].cpu().numpy(), tx, ty, W, H, bg, tile_g)
                nc_mismatch += int((res["n_contrib"][: yb - ya, : xb - xa] != nc_tile.cpu().numpy()).sum())
                got_px = r.out_color[:, ya:yb, xa:xb].cpu().numpy()
                assert np.abs(res["out"][:, : yb - ya, : xb - xa] - got_px).max() <= 1e-4 or nc_mismatch > 0
                for j_rec, gid in enumerate(ids.cpu().numpy()):
                    j = where.get(int(gid))
                    if j is None:
                        continue
                    sums["dL_dmean2D"][j] += res["d_mean"][j_rec]
                    sums["dL_dconic"][j] += res["d_conic"][j_rec]
                    sums["dL_dopacity"][j] += res["d_op"][j_rec]
                    sums["dL_dcolors"][j] += res["d_col"][j_rec]
        for k in acc:
            acc[k][(wx, wy)] = sums[k]
    assert nc_mismatch <= 8, nc_mismatch           # exp implementations may flip a threshold on a handful of pixels
    cand = np.concatenate(cand_all)
    exp = {k: np.concatenate([acc[k][w] for w in wins]) for k in acc}
    # up to 2 000 of them, chosen at random
    pick = np.random.default_rng(12).permutation(cand.size)[:2000]
    cand, exp = cand[pick], {k: v[pick] for k, v in exp.items()}
    assert cand.size >= 500, cand.size
    idx = torch.from_numpy(cand).to("cuda:0")
    got = {"dL_dmean2D": got_dev["dL_dmean2D"][idx].cpu().numpy(),
           "dL_dconic": got_dev["dL_dconic_opacity"][idx][:, :3].cpu().numpy(),
           "dL_dopacity": got_dev["dL_dconic_opacity"][idx][:, 3:4].cpu().numpy(),
           "dL_dcolors": got_dev["dL_dcolors"][idx].cpu().numpy()}
    nonzero = int((np.abs(exp["dL_dcolors"]).sum(1) > 0).sum())
    assert nonzero >= 100, nonzero
    for k in exp:
        scale = max(1e-6, float(np.abs(exp[k]).max()))
        err = np.abs(got[k] - exp[k]).max(1)
        # float32 atomics in varying order against a float64 sum: 2e-4 of the largest gradient; a threshold flip on a
        # pixel (counted above) may move one Gaussian further
        assert int((err > 2e-4 * scale).sum()) <= nc_mismatch, (k, float(err.max()), scale)
    # the chain to the inputs for 300 of them that received gradient
    with_grad = np.nonzero(np.abs(exp["dL_dconic"]).sum(1) > 0)[0][:300]
    ids = cand[with_grad]
    host_scene = scenes.scene_rows(scene, torch.from_numpy(ids).to("cuda:0"))
    full = {k: got_dev[k][torch.from_numpy(ids).to("cuda:0")].cpu().numpy() for k in
            ("dL_dconic_opacity", "dL_dmean2D", "dL_dcov3D", "dL_dmeans3D", "dL_dscales", "dL_drotations")}
    g_host = {"cov3D": geo["cov3D"][torch.from_numpy(ids).to("cuda:0")].cpu().numpy()}
    mags = check_backward_chain(full, g_host, host_scene, cam, W, H, np.arange(ids.size))
    assert all(m > 0 for m in mags)
