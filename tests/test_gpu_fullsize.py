"""GPU, BASELINE sizes: properties that need no CPU oracle run — sortedness, list/range
consistency, conservation of the (key, value) multiset, idempotence, band-sharded == whole
frame (bit for bit), linearity in the background — on the 5.8M-splat 1920x1080 workload, plus a
mid-size 1080p frame checked pixel by pixel against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from helpers import ROOT, assert_blend_parity, image_report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def garden():
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    scene = scenes.garden_like_scene(5_834_784, seed=43)
    span = float(np.max(scene["means3D"][:, :3].max(0) - scene["means3D"][:, :3].min(0)))
    cam = camera.default_camera(1920, 1080, near=0.001 * span, far=span)
    r = SplatRasterizer(1920, 1080, background=(0.0, 0.0, 0.0))
    r.configure_from_scene(scene)
    img = r.draw(cam, count_staged=True).clone()
    return torch, r, cam, img


def test_fullsize_lists_are_sorted_complete_and_ranged(garden):
    torch, r, cam, _ = garden
    g = r.map_geometry_state()
    R = r.last_num_rendered
    tt = g["tilesTouched"].to(torch.int64) & 0xFFFFFFFF
    assert int(tt.sum()) == R == (int(g["pointOffsets"][-1]) & 0xFFFFFFFF)
    assert bool(((g["radii"] > 0) >= (tt > 0)).all())
    b = r.map_binning_state()
    keys, vals = b["keys"], b["values"]
    assert bool((keys[1:] >= keys[:-1]).all()), "sorted keys out of order"
    tie = keys[1:] == keys[:-1]
    assert bool((vals[1:][tie] > vals[:-1][tie]).all()), "equal keys must keep ascending Gaussian index"
    assert r.last_plan == "blocks"          # R / N = 46 here: the lists are written directly, nothing R-sized is sorted
    # every key's depth half is the depth of its Gaussian; its tile half lies inside the grid
    depth_bits = g["depths"].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    samp = torch.randint(0, R, (2_000_000,), device=keys.device)
    assert bool(((keys[samp] & 0xFFFFFFFF) == depth_bits[vals[samp].to(torch.int64)]).all())
    tiles = keys >> 32
    assert int(tiles.max()) < 120 * 68 and int(tiles.min()) >= 0
    # ranges: [start, end) of each tile's run (GSCuda.cu:504-538)
    rg = r.map_image_state()["ranges"].to(torch.int64)
    counts = torch.bincount(tiles, minlength=120 * 68)
    assert bool(((rg[:, 1] - rg[:, 0]) == counts).all())
    starts = torch.cumsum(counts, 0) - counts
    assert bool((rg[counts > 0, 0] == starts[counts > 0]).all())


def test_fullsize_both_binning_plans_give_identical_lists(garden):
    """The block plan (sorted lists written directly) against the sort plan (column-major emission + onesweep
    pass): 267 M keys / values and all tile ranges bit for bit; under the sort plan the sort is a
    permutation of what was emitted (checksum of checksums)."""
    torch, r, cam, img = garden
    r.draw(cam, plan="blocks")
    assert r.last_plan == "blocks"
    b = r.map_binning_state()
    keys_b, vals_b = b["keys"].clone(), b["values"].clone()
    ranges_b = r.map_image_state()["ranges"].clone()
    img_s = r.draw(cam, plan="sort").clone()
    assert r.last_plan == "sort"
    b = r.map_binning_state()
    assert torch.equal(b["keys"], keys_b) and torch.equal(b["values"], vals_b)
    assert torch.equal(r.map_image_state()["ranges"], ranges_b)
    assert torch.equal(img_s, img)
    ku, vu = b["keys_unsorted"], b["values_unsorted"]
    for f in (lambda k, v: k.sum(), lambda k, v: (k ^ (v.to(torch.int64) * 0x9E3779B1)).sum(),
              lambda k, v: v.to(torch.int64).sum()):
        assert int(f(ku, vu)) == int(f(b["keys"], b["values"]))
    del keys_b, vals_b
    r.draw(cam)


def test_fullsize_frame_is_idempotent_and_band_sharding_is_bit_exact(garden):
    torch, r, cam, img = garden
    again = r.draw(cam).clone()
    assert torch.equal(img, again)
    r.out_color.fill_(-1.0)
    total_r = 0
    for rows in ((0, 9), (9, 31), (31, 32), (32, 68)):          # uneven bands, as ranks would hold them
        r.draw(cam, tile_rows=rows)
        total_r += r.last_num_rendered
    assert torch.equal(r.out_color, img), "tile-row bands must reproduce the single-GPU pixels exactly"
    r.draw(cam)
    assert total_r == r.last_num_rendered                        # bands partition the instances


def test_fullsize_background_enters_linearly(garden):
    torch, r, cam, img = garden
    ft = r.map_image_state()["finalT"].clone()
    assert float(ft.min()) >= 0.0 and float(ft.max()) <= 1.0
    nc = r.map_image_state()["nContrib"].to(torch.int64)
    rg = r.map_image_state()["ranges"].to(torch.int64)
    per_tile = (rg[:, 1] - rg[:, 0]).view(68, 120)
    limit = per_tile.repeat_interleave(16, 0).repeat_interleave(16, 1)[:1080, :1920]
    assert bool((nc <= limit).all())
    r.background.copy_(torch.tensor([0.25, 0.5, 1.0], device=r.background.device))
    img2 = r.draw(cam).clone()
    r.background.zero_()
    want = img + ft[None] * torch.tensor([0.25, 0.5, 1.0], device=ft.device)[:, None, None]
    assert float((img2 - want).abs().max()) <= 1e-6


def test_fullsize_frame_against_oracle(garden):
    """BASELINE config 2 stand-in at full size, against the CPU oracle on the same inputs (the oracle
    needs about half a minute of host time for the stable sort of 267 M keys)."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    torch, r, cam, img = garden
    scene = scenes.garden_like_scene(5_834_784, seed=43)
    exp = cpu_oracle.forward(scene, cam, (0.0, 0.0, 0.0), threads=max(1, min(64, cpu_oracle.hardware_concurrency())))
    r.background.zero_()
    img = r.draw(cam, count_staged=True).cpu().numpy()
    assert r.last_num_rendered == exp["num_rendered"] and r.last_records_staged == exp["records_staged"]
    g = r.map_geometry_state()
    # (per-Gaussian arrays are written for Gaussians with a tile only, as in the reference: what the chunk holds elsewhere
    # is whatever the allocator handed out)
    vis = exp["tilesTouched"] > 0
    assert np.array_equal(g["tilesTouched"].cpu().numpy().view(np.uint32), exp["tilesTouched"])
    assert np.array_equal(g["radii"].cpu().numpy(), exp["radii"])
    for k in ("means2D", "depths", "cov3D", "rgb", "conicOpacity"):
        assert np.array_equal(g[k].cpu().numpy()[vis], exp[k][vis]), k
    b = r.map_binning_state()
    assert np.array_equal(b["keys"].cpu().numpy().view(np.uint64), exp["keys"])
    assert np.array_equal(b["values"].cpu().numpy().view(np.uint32), exp["values"])
    assert np.array_equal(r.map_image_state()["ranges"].cpu().numpy().view(np.uint32), exp["ranges"])
    im = r.map_image_state()
    assert_blend_parity(img, im["finalT"].cpu().numpy(), im["nContrib"].cpu().numpy(), exp,
                        f"full-size frame: R={r.last_num_rendered} R_f={r.last_records_staged} plan={r.last_plan}")


def test_1080p_midsize_frame_against_oracle():
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    from oracle import cpu_oracle
    scene = scenes.garden_like_scene(150_000, seed=43)
    scene["means3D"][:, 2] += 6.0                                   # camera in front of the cloud
    cam = camera.default_camera(1920, 1080, near=0.05, far=80.0)
    exp = cpu_oracle.forward(scene, cam, (0.05, 0.05, 0.05), threads=max(1, min(32, cpu_oracle.hardware_concurrency())))
    r = SplatRasterizer(1920, 1080, background=(0.05, 0.05, 0.05))
    r.configure_from_scene(scene)
    img = r.draw(cam, count_staged=True).cpu().numpy()
    assert r.last_num_rendered == exp["num_rendered"] and r.last_records_staged == exp["records_staged"]
    b = r.map_binning_state()
    assert np.array_equal(b["keys"].cpu().numpy().view(np.uint64), exp["keys"])
    assert np.array_equal(b["values"].cpu().numpy().view(np.uint32), exp["values"])
    im = r.map_image_state()
    assert_blend_parity(img, im["finalT"].cpu().numpy(), im["nContrib"].cpu().numpy(), exp, "1080p mid-size frame")


def test_4k_frame_both_binning_plans_agree():
    """3840 x 2160 (BASELINE config 3's resolution): 240 x 135 tiles = 510 tile blocks, the largest
    block table the block plan takes (130 KB of LDS masks in coarse_emit). Lists, ranges and pixels must
    match the sort plan bit for bit."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    scene = scenes.garden_like_scene(1_000_000, seed=47)
    span = float(np.max(scene["means3D"][:, :3].max(0) - scene["means3D"][:, :3].min(0)))
    cam = camera.default_camera(3840, 2160, near=0.001 * span, far=span)
    r = SplatRasterizer(3840, 2160, background=(0.1, 0.0, 0.2))
    r.configure_from_scene(scene)
    img_b = r.draw(cam, plan="blocks").clone()
    assert r.last_plan == "blocks" and r.last_num_rendered > 50_000_000
    b = r.map_binning_state()
    keys_b, vals_b = b["keys"].clone(), b["values"].clone()
    ranges_b = r.map_image_state()["ranges"].clone()
    img_s = r.draw(cam, plan="sort")
    assert r.last_plan == "sort"
    b = r.map_binning_state()
    assert torch.equal(b["keys"], keys_b) and torch.equal(b["values"], vals_b)
    assert torch.equal(r.map_image_state()["ranges"], ranges_b)
    assert torch.equal(img_s, img_b)
    assert bool((keys_b[1:] >= keys_b[:-1]).all())


def test_cpp_harness_through_the_reference_signature(tmp_path):
    """A C++ caller using gscuda::forward(...) from include/gscuda_shim.hpp gets the golden image."""
    from helpers import load_golden
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness")])
    scene, cam, bg, exp = load_golden()
    n = scene["means3D"].shape[0]
    path_in, path_out = tmp_path / "scene.bin", tmp_path / "out.bin"
    with open(path_in, "wb") as f:
        np.array([n, cam.width, cam.height], np.int32).tofile(f)
        np.concatenate([cam.view, cam.proj, cam.cam_pos, [cam.tan_fovx, cam.tan_fovy], bg]).astype(np.float32).tofile(f)
        for k in ("means3D", "scales", "rotations", "opacities", "shs"):
            np.ascontiguousarray(scene[k], np.float32).tofile(f)
    out = subprocess.check_output([os.path.join(ROOT, "harness", "gsr_harness"), str(path_in), str(path_out), "3"], text=True)
    assert f"numRendered={exp['num_rendered']}" in out, out
    img = np.fromfile(path_out, np.float32).reshape(3, cam.height, cam.width)
    assert np.abs(img - exp["out_color"]).max() <= 1e-4


def test_overlapped_emission_gives_the_same_frame():
    """GSR_FLAG_OVERLAP_EMIT: the blend (which reads the block lists) runs on a second stream beside the emission.
    Keys, values, ranges, pixels, nContrib and the staged-record count must not change."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    scene = scenes.garden_like_scene(400_000, seed=43)
    scene["means3D"][:, 2] += 6.0
    cam = camera.default_camera(1920, 1080, near=0.05, far=80.0)
    r = SplatRasterizer(1920, 1080, background=(0.05, 0.1, 0.15))
    r.configure_from_scene(scene)
    img = r.draw(cam, plan="blocks", count_staged=True).clone()
    staged = r.last_records_staged
    b = r.map_binning_state()
    keys, vals = b["keys"].clone(), b["values"].clone()
    nc = r.map_image_state()["nContrib"].clone()
    for cb in (r.binning,):
        cb.tensor.zero_()
    for _ in range(3):
        img2 = r.draw(cam, plan="blocks", count_staged=True, overlap_emit=True)
        assert r.last_plan == "blocks" and r.last_records_staged == staged
        b = r.map_binning_state()
        assert torch.equal(img2, img) and torch.equal(b["keys"], keys) and torch.equal(b["values"], vals)
        assert torch.equal(r.map_image_state()["nContrib"], nc)
    # and the old pairing (sort plan, blend from the sorted list) agrees with both
    img3 = r.draw(cam, plan="sort", count_staged=True)
    assert torch.equal(img3, img) and r.last_records_staged == staged


def test_deep_lists_against_oracle():
    """Faint splats (opacities x 0.04): pixels stay unfinished for thousands of list positions, so the blend walks every
    tile's list to its end (the frames elsewhere in this file saturate after a few hundred records). Pixels, the last
    contributor of every pixel, the staged-record count and the lists must still be the oracle's; both binning plans'
    blends, the forward-only mode and a band-sharded call must give the same frame."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    from oracle import cpu_oracle
    scene = scenes.garden_like_scene(400_000, seed=43)
    scene["means3D"][:, 2] += 6.0
    scene["opacities"] *= np.float32(0.04)
    cam = camera.default_camera(1920, 1080, near=0.05, far=80.0)
    bg = (0.05, 0.1, 0.15)
    exp = cpu_oracle.forward(scene, cam, bg, threads=max(1, min(64, cpu_oracle.hardware_concurrency())))
    assert exp["nContrib"].max() > 3000
    r = SplatRasterizer(1920, 1080, background=bg)
    r.configure_from_scene(scene)
    img = r.draw(cam, plan="blocks", count_staged=True).clone()
    assert r.last_plan == "blocks" and r.last_num_rendered == exp["num_rendered"]
    staged_blocks = r.last_records_staged
    b = r.map_binning_state()
    assert np.array_equal(b["values"].cpu().numpy().view(np.uint32), exp["values"])
    ft = r.map_image_state()["finalT"].cpu().numpy()
    assert_blend_parity(img.cpu().numpy(), ft, r.map_image_state()["nContrib"].cpu().numpy(), exp, "blend-bound frame")
    assert ft.min() >= 0.0      # no hand-over flag left in the sign bit
    assert staged_blocks == exp["records_staged"], (staged_blocks, exp["records_staged"])
    nc0, ft0 = r.map_image_state()["nContrib"].clone(), r.map_image_state()["finalT"].clone()
    img_s = r.draw(cam, plan="sort", count_staged=True)
    assert r.last_records_staged == exp["records_staged"]
    assert torch.equal(img_s, img) and torch.equal(r.map_image_state()["nContrib"], nc0) and torch.equal(r.map_image_state()["finalT"], ft0)
    img_f = r.draw(cam, plan="blocks", sorted_lists=False)
    assert torch.equal(img_f, img)
    # and in a band-sharded call (four waves per tile)
    r.out_color.fill_(-1.0)
    for rows in ((0, 20), (20, 21), (21, 68)):
        r.draw(cam, tile_rows=rows)
    assert torch.equal(r.out_color, img)
