"""GPU: deep tiles (csrc/blend.hip, GSR_FLAG_DEEP_TILES_ALL / GSR_FLAG_NO_DEEP_TILES) — a tile composited by four waves that
walk its list together, one 16 x 4 strip each. Every pixel must see the same records in the same order as with one wave per
tile: out_color, finalT, nContrib bit for bit, and the staged-record count R_f (which the deep way derives from WHERE the tile
finished, not from counting batches). Against the ordinary way, against the oracle, on ragged sizes, bands of tile rows, the
upstream semantics, faint splats (every list walked to its end), and with the history choosing the tiles by itself."""
import numpy as np
import pytest

from helpers import assert_blend_parity, load_golden

pytestmark = pytest.mark.gpu


def _state(r):
    st = r.map_image_state()
    return r.out_color.clone(), st["finalT"].clone(), st["nContrib"].clone()


def _poison(r):
    r.out_color.fill_(float("nan"))
    st = r.map_image_state()
    st["finalT"].fill_(-7.0)
    st["nContrib"].fill_(-7)


def _same(a, b):
    import torch
    return all(torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)
               for x, y in zip(a, b))


def test_config1_by_four_waves_per_tile_against_the_golden_fixture():
    from gsrast_amd.rasterizer import SplatRasterizer
    scene, cam, bg, exp = load_golden()
    r = SplatRasterizer(cam.width, cam.height, background=bg)
    r.configure_from_scene(scene)
    for plan, mode in (("sort", "all"), ("blocks", "all"), ("sort", "all8"), ("sort", "all16")):
        # (the block plan blends a frame this sparse from the sorted lists: the same kernel)
        r.draw(cam, plan=plan, tile_history=False, deep_tiles=mode, count_staged=True)
        assert r.last_deep_tiles
        st = r.map_image_state()
        assert_blend_parity(r.out_color.cpu().numpy(), st["finalT"].cpu().numpy().reshape(cam.height, cam.width),
                            st["nContrib"].cpu().numpy().reshape(cam.height, cam.width), exp, f"config 1, deep tiles, plan {plan}", bitwise_t=False)
        assert r.last_records_staged == exp["records_staged"], (plan, r.last_records_staged, exp["records_staged"])


@pytest.mark.parametrize("size,n,pos,opacity_scale", [((640, 368), 150_000, (0.0, 0.0, -6.0), 1.0),
                                                        ((333, 217), 60_000, (0.2, -0.1, -4.0), 1.0),       # ragged: 21 x 14 tiles, last row and column partial
                                                        ((640, 368), 150_000, (0.0, 0.0, -12.0), 0.1),      # faint splats: the lists are walked to their ends
                                                        ((1280, 720), 400_000, (0.0, 0.0, -9.0), 1.0)])
def test_deep_and_ordinary_tiles_give_the_same_bits(size, n, pos, opacity_scale):
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    from oracle import cpu_oracle
    w, h = size
    scene = scenes.garden_like_scene(n, seed=77)
    scene["means3D"][:, :3] *= 0.35
    scene["opacities"] = (scene["opacities"] * opacity_scale).astype(np.float32)
    cam = camera.default_camera(w, h, near=0.05, far=60.0, position=pos)
    r = SplatRasterizer(w, h, background=(0.1, 0.2, 0.3))
    r.configure_from_scene(scene)
    r.draw(cam, plan="sort", tile_history=False, deep_tiles=False, count_staged=True)
    ref, ref_staged = _state(r), r.last_records_staged
    assert not r.last_deep_tiles
    # four waves per tile (16 x 4 strips), eight (16 x 2), sixteen (16 x 1: the upper lanes idle)
    for mode in ("all", "all8", "all16"):
        _poison(r)
        r.draw(cam, plan="sort", tile_history=False, deep_tiles=mode, count_staged=True)
        assert r.last_deep_tiles
        got, got_staged = _state(r), r.last_records_staged
        assert _same(got, ref), mode
        assert got_staged == ref_staged, (mode, got_staged, ref_staged)
    # ... and both are the oracle's
    exp = cpu_oracle.forward(scene, cam, background=(0.1, 0.2, 0.3))
    assert_blend_parity(got[0].cpu().numpy(), got[1].cpu().numpy().reshape(h, w), got[2].cpu().numpy().reshape(h, w), exp,
                        f"deep tiles {w}x{h} n={n} opacities x {opacity_scale}")
    assert got_staged == exp["records_staged"]
    # a band of tile rows (what a rank of a sharded frame renders)
    rows = (3, 11)
    band_ref = r.draw(cam, plan="sort", tile_rows=rows, tile_history=False, deep_tiles=False).clone()
    import torch
    y0, y1 = rows[0] * 16, min(rows[1] * 16, h)
    for mode in ("all", "all8", "all16"):
        r.out_color.fill_(float("nan"))
        band = r.draw(cam, plan="sort", tile_rows=rows, tile_history=False, deep_tiles=mode)
        assert torch.equal(band[:, y0:y1].view(torch.int32), band_ref[:, y0:y1].view(torch.int32)), mode


def test_deep_tiles_under_the_upstream_semantics():
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    w, h = 480, 272
    scene = scenes.garden_like_scene(80_000, seed=79)
    scene["means3D"][:, :3] *= 0.35
    cam = camera.default_camera(w, h, near=0.05, far=60.0, position=(0.0, 0.0, -5.0))
    r = SplatRasterizer(w, h, background=(0.3, 0.2, 0.1))
    r.configure_from_scene(scene)            # (SH degree 0: the DC triple leads either layout)
    r.draw(cam, plan="sort", semantics="inria", sh_degree=0, tile_history=False, deep_tiles=False, count_staged=True)
    ref, ref_staged = _state(r), r.last_records_staged
    _poison(r)
    r.draw(cam, plan="sort", semantics="inria", sh_degree=0, tile_history=False, deep_tiles="all", count_staged=True)
    assert _same(_state(r), ref) and r.last_records_staged == ref_staged


def test_sparse_frames_take_the_deep_way_by_themselves_and_nothing_changes():
    """Below 16 instances per visible Gaussian every tile is composited by four waves without anybody asking (no history
    needed: the rule is R against V), above it none is; with and without the tile history, which only orders the tiles, every
    frame is the bits of the ordinary way. Then the camera jumps between the two regimes."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    w, h = 640, 368
    scene = scenes.garden_like_scene(150_000, seed=71)
    scene["means3D"][:, :3] *= 0.35
    r = SplatRasterizer(w, h, background=(0.1, 0.2, 0.3))
    r.configure_from_scene(scene)
    cams = [camera.default_camera(w, h, near=0.05, far=60.0, position=p) for p in ((0.0, 0.0, -16.0), (0.0, 0.0, -16.4), (0.3, -0.2, -2.5))]
    refs, sparse = [], []
    for c in cams:
        r.draw(c, plan="sort", tile_history=False, deep_tiles=False, count_staged=True)
        refs.append((_state(r), r.last_records_staged))
        visible = int((r.map_geometry_state()["tilesTouched"] != 0).sum().item())
        sparse.append(r.last_num_rendered < 16 * visible)
    assert sparse[0] and sparse[1] and not sparse[2], sparse          # (far away: a tile or two per splat; close up: dozens)
    for hist in (False, True):
        for c in (0,) * 5 + (1,) * 3 + (2,) * 4 + (0,) * 3:
            _poison(r)
            r.draw(cams[c], plan="sort", tile_history=hist, count_staged=True)
            assert r.last_deep_tiles == sparse[c], (c, hist)
            assert _same(_state(r), refs[c][0]), (c, hist)
            assert r.last_records_staged == refs[c][1], (c, hist)
    assert torch.isfinite(r.out_color).all()
