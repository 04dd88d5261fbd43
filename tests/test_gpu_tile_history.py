"""GPU: the blend starts the slow tiles of the call before first (GSR_FLAG_NO_TILE_HISTORY, csrc/blend_core.hpp TileOrder).
Only WHEN a tile is composited may depend on it: every output must be the bits a call without the history gives, with fresh,
stale and foreign ticks, on a band of tile rows, after a change of size, and with the tiles' pixels filled with garbage
beforehand (a tile the order left out would otherwise show the previous frame's correct pixels)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state(r):
    st = r.map_image_state()
    return r.out_color.clone(), st["finalT"].clone(), st["nContrib"].clone()


def _draw_poisoned(r, cam, **kw):
    r.out_color.fill_(float("nan"))
    st = r.map_image_state()
    st["finalT"].fill_(-7.0)
    st["nContrib"].fill_(-7)
    r.draw(cam, **kw)
    return _state(r)


def _same(a, b):
    import torch
    return all(torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)
               for x, y in zip(a, b))


@pytest.mark.parametrize("plan,route", [("sort", True), ("blocks", True), ("blocks", "default")])
def test_reordered_tiles_give_the_same_frame(plan, route):
    """route: True = the rasterizer object's own gsr_tile_history, "default" = none passed (the library's own for the calling
    thread and stream: what a caller of the reference's signature gets)."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    w, h = 640, 368
    scene = scenes.garden_like_scene(120_000, seed=71)
    scene["means3D"][:, :3] *= 0.35
    r = SplatRasterizer(w, h, background=(0.1, 0.2, 0.3))
    r.configure_from_scene(scene)
    cams = [camera.default_camera(w, h, near=0.05, far=60.0, position=p) for p in ((0.0, 0.0, -6.0), (0.3, -0.2, -3.0), (0.0, 0.0, -12.0))]
    r.draw(cams[0], plan=plan, tile_history=False)                 # (the chunks exist from here on)
    refs = [_draw_poisoned(r, c, plan=plan, tile_history=False) for c in cams]
    assert not r.last_tiles_reordered
    assert all(bool(torch.isfinite(ref[0]).all()) for ref in refs)
    reordered = 0
    # fresh ticks (same camera several times), then stale ones (the camera jumps), then a band of tile rows
    for c, ref in ((0, refs[0]),) * 7 + ((1, refs[1]),) * 3 + ((2, refs[2]),) * 6 + ((0, refs[0]),) * 2:
        got = _draw_poisoned(r, cams[c], plan=plan, tile_history=route)
        reordered += int(r.last_tiles_reordered)
        assert _same(got, ref), (plan, c)
    # (a frame of 920 tiles cannot fill 5 120 wave slots: its longest tile always counts as slow, the order is in use)
    assert reordered >= 8, reordered
    rows = (5, 17)
    band_ref = r.draw(cams[0], plan=plan, tile_rows=rows, tile_history=False).clone()
    for _ in range(6):
        r.out_color.fill_(float("nan"))
        band = r.draw(cams[0], plan=plan, tile_rows=rows, tile_history=route)
        assert torch.equal(band[:, 16 * rows[0]:16 * rows[1]].view(torch.int32), band_ref[:, 16 * rows[0]:16 * rows[1]].view(torch.int32))
    # another size in between (with "default" the two objects share the thread's history for the stream: cleared at every change of size)
    r2 = SplatRasterizer(320, 208, background=(0.0, 0.0, 0.0))
    r2.configure_from_scene(scene)
    cam2 = camera.default_camera(320, 208, near=0.05, far=60.0, position=(0.0, 0.0, -6.0))
    r2.draw(cam2, tile_history=False)
    ref2 = _draw_poisoned(r2, cam2, tile_history=False)
    for _ in range(5):
        assert _same(_draw_poisoned(r2, cam2, tile_history=route), ref2)
        assert _same(_draw_poisoned(r, cams[0], plan=plan, tile_history=route), refs[0])


def _two_views(w, h):
    from gsrast_amd import camera
    return [camera.default_camera(w, h, near=0.05, far=60.0, position=p) for p in ((0.0, 0.0, -2.5), (0.4, 0.3, -14.0))]


def test_two_views_alternating_on_one_history_drop_the_order_and_two_histories_keep_it():
    """Same size, unrelated views, every call the other one (a stereo pair, a trainer): on ONE history each frame would be
    ordered by the other view's tile times. The library notices (the two last frames do not resemble each other), goes back
    to the patch order and says so in plan_used; with a history per view (one per rasterizer object) every view keeps its
    order. The frames are the same bits either way."""
    import torch
    from gsrast_amd import scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    w, h = 640, 368
    scene = scenes.garden_like_scene(120_000, seed=73)
    scene["means3D"][:, :3] *= 0.35
    cams = _two_views(w, h)
    one = SplatRasterizer(w, h, background=(0.1, 0.2, 0.3))
    one.configure_from_scene(scene)
    one.draw(cams[0], tile_history=False)
    refs = [_draw_poisoned(one, c, tile_history=False) for c in cams]
    assert not _same(refs[0], refs[1])
    dropped = []
    for i in range(16):
        assert _same(_draw_poisoned(one, cams[i % 2]), refs[i % 2]), i
        dropped.append(one.last_tile_order_dropped)
    assert any(dropped[:10]) and all(dropped[10:]), dropped
    # the camera settles on one view: within a few frames the order is back
    for i in range(10):
        assert _same(_draw_poisoned(one, cams[0]), refs[0])
    assert not one.last_tile_order_dropped and one.last_tiles_reordered
    # one history per view
    pair = [SplatRasterizer(w, h, background=(0.1, 0.2, 0.3)) for _ in cams]
    for r, c in zip(pair, cams):
        r.configure_from_scene(scene)
        r.draw(c, tile_history=False)                       # (the chunks exist from here on)
    seen = [False, False]
    for i in range(16):
        r = pair[i % 2]
        assert _same(_draw_poisoned(r, cams[i % 2]), refs[i % 2]), i
        assert not r.last_tile_order_dropped
        seen[i % 2] = seen[i % 2] or r.last_tiles_reordered
    assert all(seen)


def test_calls_on_two_streams_of_one_thread_do_not_share_a_history():
    """Two rasterizers of one host thread, each on a stream of its own, no synchronisation between their calls (double
    buffering): the library's own histories are kept per stream, so the second call's sort of the order never rewrites what
    the first call's blend is still reading. A caller's own history taken from one stream to the other is ordered behind
    the old stream's work by the library."""
    import torch
    from gsrast_amd import scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    w, h = 960, 544
    scene = scenes.garden_like_scene(400_000, seed=74)
    scene["means3D"][:, :3] *= 0.35
    cams = _two_views(w, h)
    rs = [SplatRasterizer(w, h, background=(0.1, 0.2, 0.3)) for _ in range(2)]
    refs = []
    for r, c in zip(rs, cams):
        r.configure_from_scene(scene)
        r.draw(c, tile_history=False)
        refs.append(_draw_poisoned(r, c, tile_history=False))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for route in ("default", "own", "swapped"):
        for i in range(10):
            for k in (0, 1):
                r = rs[k]
                r.out_color.fill_(float("nan"))
            torch.cuda.synchronize()
            for k in (0, 1):
                r = rs[k]
                # "swapped": both rasterizers pass rasterizer 0's history, from two streams in turn
                if route == "swapped":
                    saved, r._history = r._history, rs[0]._history
                with torch.cuda.stream(streams[k]):
                    r.draw(cams[k], sync=False, tile_history="default" if route == "default" else True)
                if route == "swapped":
                    r._history = saved
            torch.cuda.synchronize()
            for k in (0, 1):
                rs[k].poll_async_error()
                assert _same(_state(rs[k]), refs[k]), (route, i, k)


def test_backward_after_a_reordered_frame_is_unchanged():
    """gsr_backward reads what the forward call left (finalT, nContrib, the lists): none of it depends on the order of the tiles."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    w, h = 480, 272
    scene = scenes.garden_like_scene(60_000, seed=72)
    scene["means3D"][:, :3] *= 0.3
    r = SplatRasterizer(w, h, background=(0.1, 0.2, 0.3))
    r.configure_from_scene(scene)
    cam = camera.default_camera(w, h, near=0.05, far=50.0)
    dL = torch.randn((3, h, w), generator=torch.Generator().manual_seed(9)).cuda()
    r.draw(cam, tile_history=False)
    ref = {k: v.clone() for k, v in r.backward(dL).items()}
    seen = False
    for _ in range(6):
        r.draw(cam)
        seen = seen or r.last_tiles_reordered
        got = r.backward(dL)
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
    assert seen


@pytest.mark.parametrize("where", ["1", "2", "2 and 37 % early", "2 and 100 % early"])
def test_colours_written_beside_the_depth_sort_are_the_preprocess_kernels(where, library_env):
    """geomState.rgb by a kernel of its own on the library's second stream (GSR_PLAN_COLORS_BESIDE) against the preprocess
    kernel writing it (GSR_FLAG_SERIAL_EMIT): the same bits for every Gaussian, zeros for the culled ones, with and without
    instances (R == 0 returns before the blend: the caller's stream must have waited all the same), on a band, and the
    picture / the backward's colour gradient built on them unchanged. where = "2": beside the BLEND, which then takes a
    record's colour straight from the SH array (the library's choice beyond 16 M Gaussians; forced here through the
    environment, as is "1" — beside the depth sort, its choice at this size); "... early": a share of the Gaussians gets its
    colours right behind the preprocess, the rest beside the blend (what the library does where the history says the colours
    kernel would outlast the blend)."""
    import torch
    if "early" in where:
        library_env(GSR_COLORS_BESIDE="2", GSR_COLORS_EARLY_PCT=where.split()[2])
    else:
        library_env(GSR_COLORS_BESIDE=where, GSR_COLORS_EARLY_PCT="0")
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    w, h = 512, 288
    scene = scenes.garden_like_scene(90_000, seed=81)
    scene["means3D"][:, :3] *= 0.3
    r = SplatRasterizer(w, h, background=(0.1, 0.2, 0.3))
    r.configure_from_scene(scene)
    cam = camera.default_camera(w, h, near=0.05, far=50.0)
    # (deep tiles: the four waves of a tile take the records' colours from the SH array too, where = "2")
    for kw in ({}, {"plan": "sort"}, {"plan": "sort", "deep_tiles": "all"}, {"plan": "sort", "deep_tiles": "all8"},
               {"plan": "blocks", "overlap_emit": True}, {"tile_rows": (3, 11)}):
        img0 = r.draw(cam, **{**kw, "overlap_emit": False}).clone()
        assert not r.last_colors_beside
        rgb0 = r.map_geometry_state()["rgb"].clone()
        r.map_geometry_state()["rgb"].fill_(float("nan"))
        img1 = r.draw(cam, **kw).clone()
        assert r.last_colors_beside
        rgb1 = r.map_geometry_state()["rgb"]
        assert torch.equal(rgb0.view(torch.int32), rgb1.view(torch.int32))
        assert torch.equal(img0.view(torch.int32), img1.view(torch.int32))
    vis = r.map_geometry_state()["tilesTouched"] != 0
    assert bool((rgb1[~vis] == 0).all()) and bool((rgb1[vis] != 0).any())
    # a frame without instances: the camera looks away; the colours (all zero: nothing has a tile) are there when the stream is
    away = camera.default_camera(w, h, near=0.05, far=50.0, position=(0.0, 0.0, 500.0))
    r.map_geometry_state()["rgb"].fill_(float("nan"))
    r.draw(away)
    assert r.last_num_rendered == 0
    assert bool((r.map_geometry_state()["rgb"] == 0).all())
