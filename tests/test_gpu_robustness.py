"""GPU: the boundary's error behaviour and odd-but-legal uses of gsr_forward."""
import ctypes as C

import numpy as np
import pytest

from helpers import load_golden, single_gaussian_scene

pytestmark = pytest.mark.gpu


def _rast(w, h, **kw):
    from gsrast_amd.rasterizer import SplatRasterizer
    return SplatRasterizer(w, h, **kw)


def test_allocator_returning_null_is_reported_not_crashed():
    from gsrast_amd import _capi, camera
    scene, cam, bg, _ = load_golden()
    r = _rast(cam.width, cam.height)
    r.configure_from_scene(scene)
    r.draw(cam)                                          # healthy frame first
    r.binning._cb = _capi.ALLOC_FN(lambda user, n: None)  # the binning allocator now fails
    with pytest.raises(_capi.GsrError) as e:
        r.draw(cam)
    assert e.value.code == _capi.GSR_ERR_ALLOC
    assert r.lib.gsr_last_error() == _capi.GSR_ERR_ALLOC


def test_invalid_arguments_are_rejected_before_any_launch():
    from gsrast_amd import _capi
    L = _capi.lib()
    a = _capi.ForwardArgs()
    a.struct_size = C.sizeof(_capi.ForwardArgs)
    a.num_gaussians, a.width, a.height = 10, 64, 64
    assert L.gsr_forward(C.byref(a)) == _capi.GSR_ERR_INVALID_ARG           # no pointers at all
    scene, cam, bg, _ = load_golden()
    r = _rast(cam.width, cam.height)
    r.configure_from_scene(scene)
    with pytest.raises(_capi.GsrError):
        r.draw(cam, tile_rows=(5, 3))                                       # begin > end
    with pytest.raises(_capi.GsrError):
        r.draw(cam, tile_rows=(0, 99))                                      # beyond the grid
    r.draw(cam)                                                             # still usable afterwards


def test_tile_history_handles_are_checked_and_their_statistics_readable():
    """gsr_tile_history_*: a handle that is not one (no magic) is refused before anything is launched, NULL means the
    library's own, statistics are host-side and start empty, destroy(NULL) is a no-op."""
    import torch
    from gsrast_amd import _capi
    L = _capi.lib()
    assert L.gsr_tile_history_destroy(None) == _capi.GSR_OK
    assert L.gsr_tile_history_stats(None, (C.c_uint32 * 6)()) == _capi.GSR_ERR_INVALID_ARG
    assert L.gsr_tile_history_create(None) == _capi.GSR_ERR_INVALID_ARG
    scene, cam, bg, _ = load_golden()
    r = _rast(cam.width, cam.height)
    r.configure_from_scene(scene)
    st = r.tile_history_stats()
    assert st["calls"] == 0 and st["mean_ticks"] == 0 and not st["order_dropped"]
    ref = r.draw(cam, tile_history=False).clone()
    for _ in range(3):
        assert torch.equal(r.draw(cam), ref)
    assert r.tile_history_stats()["calls"] == 3
    # not a history: 512 zero bytes where the handle should be
    fake = (C.c_uint8 * 512)()
    good, r._history = r._history, C.c_void_p(C.addressof(fake))
    with pytest.raises(_capi.GsrError) as e:
        r.draw(cam)
    assert e.value.code == _capi.GSR_ERR_INVALID_ARG
    r._history = good
    assert torch.equal(r.draw(cam), ref)                                    # still usable afterwards
    # a second handle, created and destroyed around one frame
    h = C.c_void_p()
    assert L.gsr_tile_history_create(C.byref(h)) == _capi.GSR_OK and h.value
    r._history = h
    assert torch.equal(r.draw(cam), ref)
    torch.cuda.synchronize()
    r._history = good
    assert L.gsr_tile_history_destroy(h) == _capi.GSR_OK


def test_runs_on_a_non_default_stream_and_interleaves_two_rasterizers():
    import torch
    from gsrast_amd import camera, scenes
    scene, cam, bg, exp = load_golden()
    a, b = _rast(cam.width, cam.height), _rast(cam.width, cam.height)
    a.configure_from_scene(scene)
    other = scenes.isotropic_scene(700, seed=9)
    b.configure_from_scene(other)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ia = a.draw(cam).clone()
            ib = b.draw(cam).clone()
    s.synchronize()
    assert np.abs(ia.cpu().numpy() - exp["out_color"]).max() <= 1e-4
    ib2 = b.draw(cam)
    assert torch.equal(ib, ib2)


def test_chunks_grow_only_and_survive_resolution_independent_reuse():
    """A second, larger scene through the same rasterizer object re-allocates the chunks (2x rule of
    the reference's resizeFunctional) and still renders correctly."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    cam = camera.default_camera(160, 96)
    r = _rast(160, 96)
    for n, seed in ((200, 1), (5000, 2), (300, 3)):
        scene = scenes.isotropic_scene(n, seed=seed)
        r.configure_from_scene(scene)
        img = r.draw(cam).cpu().numpy()
        exp = cpu_oracle.forward(scene, cam)
        assert r.last_num_rendered == exp["num_rendered"]
        assert np.abs(img - exp["out_color"]).max() <= 1e-4
    caps = [r.geom.capacity, r.binning.capacity, r.image.capacity]
    assert all(c > 0 for c in caps)


def test_degenerate_inputs_do_not_hang_or_fault():
    """Zero scales, zero quaternions, NaN positions, opacity > 1: garbage in, but a finished frame out."""
    from gsrast_amd import camera
    scene = single_gaussian_scene(pos=(0, 0, 0), scale=0.2, opacity=0.7, n=64)
    scene["means3D"][:, 0] = np.linspace(-1, 1, 64)
    scene["scales"][1, :3] = 0.0
    scene["rotations"][2] = 0.0
    scene["means3D"][3, :3] = np.nan
    scene["opacities"][4] = 7.5
    scene["means3D"][5, :3] = 1e30
    scene["scales"][6, :3] = 1e6
    cam = camera.default_camera(128, 64)
    r = _rast(128, 64)
    r.configure_from_scene(scene)
    img = r.draw(cam).cpu().numpy()
    assert img.shape == (3, 64, 128)
    ok = np.ones(64, bool)
    ok[[1, 2, 3, 4, 5, 6]] = False
    g = r.map_geometry_state()
    assert int((g["radii"].cpu().numpy()[ok] > 0).sum()) > 30          # the healthy ones still render


def test_precomputed_colors_and_cov3d_paths():
    """colors_precomp / cov3D_precomp are honoured like the reference does (GSCuda.cu:315-323,362,803)."""
    import torch
    from gsrast_amd import _capi
    from oracle import cpu_oracle
    scene, cam, bg, exp = load_golden()
    r = _rast(cam.width, cam.height)
    r.configure_from_scene(scene)
    r.draw(cam)
    g = r.map_geometry_state()
    cov = g["cov3D"].clone()
    colors = torch.rand((r.num_gaussians, 3), device=cov.device)
    # second rasterizer fed with the first one's covariances and arbitrary colours
    r2 = _rast(cam.width, cam.height)
    r2.configure_from_scene(scene)
    r2.set_camera(cam)
    a = _capi.ForwardArgs()
    a.struct_size = C.sizeof(_capi.ForwardArgs)
    a.geometry_alloc, a.binning_alloc, a.image_alloc = r2.geom.callback, r2.binning.callback, r2.image.callback
    a.num_gaussians, a.sh_dims, a.M = r2.num_gaussians, 3, 16
    a.background = r2.background.data_ptr()
    a.width, a.height = cam.width, cam.height
    a.means3D, a.shs, a.colors_precomp = r2.means3D.data_ptr(), None, colors.data_ptr()
    a.opacities, a.scales, a.rotations = r2.opacities.data_ptr(), None, None
    a.scale_modifier = 1.0
    a.cov3D_precomp = cov.data_ptr()
    a.view_matrix, a.proj_matrix, a.cam_pos = r2._view.data_ptr(), r2._proj.data_ptr(), r2._cam_pos.data_ptr()
    a.tan_fovx, a.tan_fovy = r2._tan
    a.out_color = r2.out_color.data_ptr()
    a.rects = r2.rects.data_ptr()
    a.stream = torch.cuda.current_stream().cuda_stream
    _capi.check(r2.lib.gsr_forward(C.byref(a)), "gsr_forward")
    torch.cuda.synchronize()
    assert int(a.num_rendered) == exp["num_rendered"]
    # same geometry, so the same per-pixel weights: out = sum_i w_i colour_i with the weights of the DC render
    assert torch.equal(r2.map_image_state()["nContrib"], r.map_image_state()["nContrib"])
    assert torch.equal(r2.map_image_state()["finalT"], r.map_image_state()["finalT"])


def test_colors_precomputed_once_per_scene_give_the_same_frame_bit_for_bit():
    """The reference's colour is view-independent (0.5 + 0.4 DC, GSCuda.cu:362-366), so a caller may compute it once
    (gsr_colors_from_dc) and pass it as colorsPrecomp (GSCuda.cuh:111, :803): bit-equal to geomState.rgb of the `shs` call
    for every visible Gaussian, same pixels / finalT / nContrib / lists, same gradients — under every plan."""
    import torch
    from gsrast_amd import camera, scenes
    scene = scenes.garden_like_scene(50_000, seed=52)
    scene["means3D"][:, :3] *= 0.3
    W, H = 640, 368
    cam = camera.default_camera(W, H, near=0.05, far=60.0)
    r = _rast(W, H, background=(0.05, 0.1, 0.2))
    r.configure_from_scene(scene)
    dl = torch.from_numpy(np.random.default_rng(3).normal(size=(3, H, W)).astype(np.float32)).cuda()
    for plan, lists in (("sort", True), ("blocks", True), ("blocks", False)):
        ref = r.draw(cam, plan=plan, sorted_lists=lists).clone()
        assert r.last_num_rendered > 0
        g = r.map_geometry_state()
        vis = g["radii"] > 0
        rgb = g["rgb"].clone()
        im = {k: v.clone() for k, v in r.map_image_state().items()}
        keys = r.map_binning_state()["keys"].clone() if lists else None
        grads = {k: v.clone() for k, v in r.backward(dl).items()}
        g["rgb"].fill_(float("nan"))                         # the precomp call must neither write nor read geomState.rgb
        got = r.draw(cam, plan=plan, sorted_lists=lists, colors_precomp=True)
        assert torch.equal(r.precomputed_colors()[vis], rgb[vis]), plan
        assert torch.equal(got, ref), plan
        assert bool(torch.isnan(r.map_geometry_state()["rgb"]).all())
        for k, v in r.map_image_state().items():
            assert torch.equal(v, im[k]), (plan, k)
        if lists:
            assert torch.equal(r.map_binning_state()["keys"], keys)
        out = r.backward(dl)
        # (the render backward's sums and the colour chain; the covariance chain amplifies the order-dependent rounding of the
        # float atomics behind it by orders of magnitude for splats next to the camera, DESIGN.md §8 f-3, and is not compared)
        for k in ("dL_dmean2D", "dL_dconic_opacity", "dL_dcolors", "dL_dshs"):
            v = grads[k]
            scale = float(v.abs().max()) + 1e-30
            assert float((out[k] - v).abs().max()) <= 2e-4 * scale, (plan, k)


def test_overlapped_emission_on_a_non_default_stream_with_bands_and_upstream_semantics():
    """GSR_FLAG_OVERLAP_EMIT forks to the library's own side stream and joins the caller's stream again: the
    results must be those of the serial call whatever stream the caller uses, for band-limited calls and for
    the upstream semantics profile (forced block plan on a small frame)."""
    import torch
    from gsrast_amd import camera, scenes
    scene = scenes.garden_like_scene(20_000, seed=31)
    scene["means3D"][:, :3] *= 0.25
    cam = camera.default_camera(640, 360, near=0.05, far=50.0)
    r = _rast(640, 360, background=(0.2, 0.1, 0.0))
    r.configure_from_scene(scene)
    ref = r.draw(cam, plan="blocks", count_staged=True).clone()
    ref_keys = r.map_binning_state()["keys"].clone()
    staged = r.last_records_staged
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(4):
            img = r.draw(cam, plan="blocks", overlap_emit=True, count_staged=True)
            assert torch.equal(img, ref) and r.last_records_staged == staged
            assert torch.equal(r.map_binning_state()["keys"], ref_keys)
        r.out_color.fill_(-1.0)
        for rows in ((0, 5), (5, 6), (6, 23)):
            r.draw(cam, plan="blocks", overlap_emit=True, tile_rows=rows)
        assert torch.equal(r.out_color, ref)
        a = r.draw(cam, plan="blocks", semantics="inria", sh_degree=0).clone()
        b = r.draw(cam, plan="blocks", semantics="inria", sh_degree=0, overlap_emit=True)
        c = r.draw(cam, plan="sort", semantics="inria", sh_degree=0)
        assert torch.equal(a, b) and torch.equal(a, c)


def test_more_than_u32_instances_is_refused_before_the_binning_chunk():
    """The reference's offsets are u32 (AuxBuffer.cuh:51). 140 000 splats that each cover all 32 400 tiles of a 4K
    frame sum to 4.5 G instances: the u32 scan wraps to a small number, the un-wrapped 64-bit total says so, and the
    call must stop with GSR_ERR_TOO_LARGE without asking for a binning chunk sized by the wrapped count."""
    from gsrast_amd import _capi, camera
    n = 140_000
    scene = single_gaussian_scene(pos=(0.0, 0.0, 0.0), scale=30.0, opacity=0.01, n=n)
    cam = camera.default_camera(3840, 2160)
    r = _rast(3840, 2160)
    r.configure_from_scene(scene)
    with pytest.raises(_capi.GsrError) as e:
        r.draw(cam)
    assert e.value.code == _capi.GSR_ERR_TOO_LARGE
    assert r.binning.calls == []                                  # never asked for
    tt = r.map_geometry_state()["tilesTouched"].cpu().numpy().view(np.uint32).astype(np.int64)
    assert int(tt.sum()) == n * 240 * 135 > 2 ** 32
    # one splat fewer than the limit's worth is still a legal frame for the scan (not rendered here: 51 GB of lists)
    small = single_gaussian_scene(pos=(0.0, 0.0, 0.0), scale=30.0, opacity=0.01, n=100)
    r2 = _rast(3840, 2160)
    r2.configure_from_scene(small)
    r2.draw(cam)
    assert r2.last_num_rendered == 100 * 240 * 135


def test_forward_only_calls_may_skip_the_sorted_lists():
    """GSR_FLAG_NO_SORTED_LISTS under the block plan: same pixels, ranges, finalT, nContrib, R and R_f; keys / values
    are left unwritten; gsr_backward then walks the block lists instead and gives the gradients it gives from the
    sorted lists; the flag does nothing under the sort plan."""
    import torch
    from gsrast_amd import _capi, camera, scenes
    scene = scenes.garden_like_scene(300_000, seed=43)
    scene["means3D"][:, 2] += 6.0
    cam = camera.default_camera(1920, 1080, near=0.05, far=80.0)
    r = _rast(1920, 1080, background=(0.05, 0.1, 0.15))
    r.configure_from_scene(scene)
    img = r.draw(cam, plan="blocks", count_staged=True).clone()
    staged, R = r.last_records_staged, r.last_num_rendered
    assert r.last_lists_written
    dl = torch.from_numpy(np.random.default_rng(5).normal(size=(3, 1080, 1920)).astype(np.float32))
    ref = {k: v.clone() for k, v in r.backward(dl).items() if isinstance(v, torch.Tensor)}
    again = {k: v.clone() for k, v in r.backward(dl).items() if isinstance(v, torch.Tensor)}
    noise = {k: float((again[k] - v).abs().max()) for k, v in ref.items()}     # float atomics: run-to-run differences
    st = {k: v.clone() for k, v in r.map_image_state().items()}
    b = r.map_binning_state()
    b["keys"].fill_(-1); b["values"].fill_(-1)
    img2 = r.draw(cam, plan="blocks", count_staged=True, sorted_lists=False)
    assert r.last_plan == "blocks" and not r.last_lists_written
    assert torch.equal(img2, img) and r.last_records_staged == staged and r.last_num_rendered == R
    for k, v in r.map_image_state().items():
        assert torch.equal(v, st[k]), k
    b = r.map_binning_state()
    assert bool((b["keys"] == -1).all()) and bool((b["values"] == -1).all())      # nothing wrote them
    # backward from that state: the tile lists come from the block lists (same records, same order; the float atomics
    # of different tiles land in a different order, so equality is up to what two runs of one path differ by)
    got = r.backward(dl)
    b = r.map_binning_state()
    assert bool((b["keys"] == -1).all()) and bool((b["values"] == -1).all())      # still nobody wrote them
    assert set(ref) <= set(got)
    for k, v in ref.items():
        scale = float(v.abs().max()) + 1e-30
        assert float((got[k] - v).abs().max()) <= max(2e-4 * scale, 4.0 * noise[k]), (k, float((got[k] - v).abs().max()), scale, noise[k])
        assert float(v.abs().sum()) > 0, k
    # under the sort plan the blend reads the sorted list: the flag is ignored
    img3 = r.draw(cam, plan="sort", sorted_lists=False)
    assert r.last_plan == "sort" and r.last_lists_written and torch.equal(img3, img)
    r.backward(torch.ones((3, 1080, 1920)))                                        # and backward works again


def test_plan_and_blend_feed_follow_the_instances_per_visible_gaussian():
    """The default choices (csrc/api.hip): the block plan from 6 instances per VISIBLE Gaussian up, or with an eighth of the
    instances in splats of 256 tiles and more; under it a blend that does not run beside the emission reads the sorted lists
    on sparse frames (fewer than 48 per visible Gaussian) and the block lists on dense ones. Whatever is chosen, pixels,
    finalT, nContrib, ranges, R and R_f are those of every other choice."""
    import torch
    from gsrast_amd import camera, scenes
    scene = scenes.garden_like_scene(400_000, seed=47)
    r = _rast(1920, 1080, background=(0.1, 0.0, 0.2))
    r.configure_from_scene(scene)
    seen = set()
    for z in (-5.0, -14.0, -30.0, -50.0, -90.0):
        cam = camera.default_camera(1920, 1080, near=0.05, far=120.0, position=(0.0, 0.0, z))
        img = r.draw(cam, count_staged=True).clone()
        R, staged, plan, from_lists = r.last_num_rendered, r.last_records_staged, r.last_plan, r.last_blend_from_lists
        tt = r.map_geometry_state()["tilesTouched"].to(torch.int64)
        V, big = int((tt != 0).sum().item()), int(tt[tt >= 256].sum().item())
        assert plan == ("blocks" if (R >= 6 * V or 8 * big >= R) else "sort"), (z, R, V, big, plan)
        assert from_lists == (plan == "blocks" and R < 48 * V and not r.last_emit_overlapped), (z, R, V, from_lists)
        seen.add((plan, from_lists))
        st = {k: v.clone() for k, v in r.map_image_state().items()}
        for kw in (dict(plan="sort"), dict(plan="blocks"), dict(plan="blocks", sorted_lists=False)):
            other = r.draw(cam, count_staged=True, **kw)
            assert torch.equal(other, img), (z, kw)
            assert r.last_num_rendered == R and r.last_records_staged == staged, (z, kw)
            for k, v in r.map_image_state().items():
                assert torch.equal(v, st[k]), (z, kw, k)
            # without the staged-record count the blend from the sorted lists may give a tile four waves (one 16 x 4 strip
            # each) when few tiles have a list — the far poses here: same pixels, finalT, nContrib
            other = r.draw(cam, **kw)
            assert torch.equal(other, img), (z, kw, "uncounted")
            for k, v in r.map_image_state().items():
                assert torch.equal(v, st[k]), (z, kw, k, "uncounted")
    assert len(seen) == 3, seen          # the poses cover: sort plan, block plan fed from the sorted lists, block plan fed from the block lists


def test_nan_positions_of_either_sign_match_the_oracle():
    """A NaN position passes the reference's frustum test (every comparison is false, GSCuda.cu:306-309) and dies
    at the rectangle (an int conversion of NaN gives an empty one): nothing of it may reach the lists. (A visible key
    whose depth is a NaN can only come from overflow, 0 * inf, whose NaN sign is the platform's: no parity to test;
    the depth sort counts all 256 top-byte digits so that such a key is still ordered by all four passes.)"""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    scene = scenes.isotropic_scene(600, seed=21)
    cam = camera.default_camera(96, 96)
    r = _rast(96, 96)
    for sign in (1.0, -1.0):
        sc = {k: v.copy() for k, v in scene.items()}
        sc["means3D"][7, 2] = np.float32(np.nan) * np.float32(sign)
        exp = cpu_oracle.forward(sc, cam)
        r.configure_from_scene(sc)
        img = r.draw(cam).cpu().numpy()
        assert r.last_num_rendered == exp["num_rendered"]
        b = r.map_binning_state()
        assert np.array_equal(b["keys"].cpu().numpy().view(np.uint64), exp["keys"])
        assert np.array_equal(b["values"].cpu().numpy().view(np.uint32), exp["values"])
        assert not (b["values"].cpu().numpy() == 7).any()
        assert np.abs(img - exp["out_color"]).max() <= 1e-4


def test_the_librarys_rccl_communicator_comes_up_on_one_rank():
    """gsr_exchange_* load RCCL at run time (the copy PyTorch has mapped) and call it through hand-declared prototypes
    (ncclUniqueId is 128 bytes BY VALUE): on the one-GPU box a communicator of one rank is all that can be created, which
    still exercises dlopen, ncclGetUniqueId, ncclCommInitRank, the band validation of gsr_exchange_bands and
    ncclCommDestroy. (The transfers themselves need peers: their plan is checked on the CPU, tests/test_sharding_gloo.py.)"""
    import torch
    from gsrast_amd import _capi, sharding
    L = _capi.lib()
    path = sharding._torch_rccl_path()
    cpath = path.encode() if path else None
    ident = C.create_string_buffer(128)
    rc = L.gsr_exchange_unique_id(cpath, ident)
    assert rc == _capi.GSR_OK, L.gsr_exchange_last_error().decode()
    assert any(b != 0 for b in ident.raw)
    handle = C.c_void_p()
    with torch.cuda.device(0):
        rc = L.gsr_exchange_create(cpath, ident.raw, 0, 1, C.byref(handle))
    assert rc == _capi.GSR_OK and handle.value, L.gsr_exchange_last_error().decode()
    W, H = 320, 200
    frame = torch.rand((3, H, W), device="cuda")
    keep = frame.clone()
    bounds = (C.c_int32 * 2)(0, (H + 15) // 16)
    assert L.gsr_exchange_bands(handle, frame.data_ptr(), W, H, bounds, -1, None) == _capi.GSR_OK
    bad = (C.c_int32 * 2)(0, 5)
    assert L.gsr_exchange_bands(handle, frame.data_ptr(), W, H, bad, -1, None) == _capi.GSR_ERR_INVALID_ARG
    torch.cuda.synchronize()
    assert torch.equal(frame, keep)
    # the transfers themselves, as far as one GPU goes: a band's three pieces sent to the own rank (ncclGroupStart,
    # ncclSend, ncclRecv, ncclGroupEnd through the hand-declared prototypes, on a stream that is not the default one)
    side = torch.cuda.Stream()
    y0, y1 = 48, 160
    src = frame[:, y0:y1, :].contiguous()
    dst = torch.full_like(src, -1.0)
    torch.cuda.synchronize()
    rc = L.gsr_exchange_loopback(handle, src.data_ptr(), dst.data_ptr(), (y1 - y0) * W, 3, side.cuda_stream)
    assert rc == _capi.GSR_OK, L.gsr_exchange_last_error().decode()
    side.synchronize()
    assert torch.equal(dst, src)
    assert L.gsr_exchange_loopback(handle, None, dst.data_ptr(), 16, 1, None) == _capi.GSR_ERR_INVALID_ARG
    assert L.gsr_exchange_destroy(handle) == _capi.GSR_OK


def test_two_host_threads_render_at_the_same_time():
    """Two host threads, a rasterizer and a stream each, no synchronisation between them (a viewer with two frames in flight; the
    library's pinned words, events, second stream and default tile histories are per host thread and device): every frame of
    either thread — image, finalT, nContrib, the sorted lists' checksum — is bit for bit the frame the same rasterizer renders
    alone."""
    import threading
    import torch
    from gsrast_amd import camera, scenes
    W, H = 640, 368
    scene = scenes.garden_like_scene(150_000, seed=77)
    scene["means3D"][:, :3] *= 0.5
    poses = [[(0.0, 0.0, -3.0 - 0.15 * i) for i in range(12)], [(0.3, -0.2, -5.0 + 0.1 * i) for i in range(12)]]
    rasts = [_rast(W, H), _rast(W, H)]
    for r in rasts:
        r.configure_from_scene(scene)

    def frame_state(r):
        st = r.map_image_state()
        b = r.map_binning_state()
        return (r.out_color.clone(), st["finalT"].clone(), st["nContrib"].clone(), int(r.last_num_rendered),
                int(b["keys"].view(torch.int64).sum().item()), int(b["values"].to(torch.int64).sum().item()))

    alone = [[], []]
    for t in range(2):
        for p in poses[t]:
            rasts[t].draw(camera.default_camera(W, H, near=0.05, far=60.0, position=p), tile_history="default")
            alone[t].append(frame_state(rasts[t]))
    together, errors = [[], []], []

    def work(t):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for rep in range(3):
                    got = []
                    for p in poses[t]:
                        rasts[t].draw(camera.default_camera(W, H, near=0.05, far=60.0, position=p), tile_history="default")
                        got.append(frame_state(rasts[t]))
                    together[t] = got
        except Exception as e:                              # noqa: BLE001 (reported by the main thread)
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for t in range(2):
        assert len(together[t]) == len(alone[t])
        for i, (g, e) in enumerate(zip(together[t], alone[t])):
            assert g[3:] == e[3:], (t, i, g[3:], e[3:])
            for a, b in zip(g[:3], e[:3]):
                assert torch.equal(a, b), (t, i)


def test_short_lived_threads_leave_nothing_behind():
    """200 host threads, one after the other, each renders two frames the way a caller of the reference's signature does (no
    tile history of its own: the library then keeps one per host thread, device and stream — 384 KB of device memory —, a
    second stream, events, pinned words) and ends. The reference owns nothing (GSCuda.cu:723-784); here everything a thread
    made the library allocate goes back when the thread ends (gsr_thread_release): the device's free memory stays flat, and
    every frame is the frame the main thread renders. gsr_thread_release by hand, then another call: starts from nothing."""
    import threading
    import torch
    from gsrast_amd import _capi, camera, scenes
    W, H = 320, 192
    scene = scenes.garden_like_scene(20_000, seed=5)
    scene["means3D"][:, :3] *= 0.4
    r = _rast(W, H)
    r.configure_from_scene(scene)
    cam = camera.default_camera(W, H, near=0.05, far=60.0, position=(0.0, 0.0, -4.0))
    ref = r.draw(cam, tile_history="default").clone()
    errors = []

    def work():
        try:
            for _ in range(2):
                img = r.draw(cam, tile_history="default")
                if not torch.equal(img.view(torch.int32), ref.view(torch.int32)):
                    errors.append("frame differs")
        except Exception as e:                              # noqa: BLE001 (reported by the main thread)
            errors.append(repr(e))

    def burst(n):
        for _ in range(n):
            th = threading.Thread(target=work)
            th.start()
            th.join()

    burst(8)                                                # (whatever the runtimes set up per process is set up)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    burst(200)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert not errors, errors[:3]
    # (200 histories alone would be 77 MB)
    assert free0 - free1 < 8 * 1024 * 1024, (free0, free1)
    L = _capi.lib()
    assert L.gsr_thread_release() == _capi.GSR_OK
    img = r.draw(cam, tile_history="default")
    assert torch.equal(img.view(torch.int32), ref.view(torch.int32))
    assert L.gsr_thread_release() == _capi.GSR_OK
