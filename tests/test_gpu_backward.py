"""GPU: gsr_backward (render backward + covariance / colour chain) against the float64 gradient oracle
(oracle/backward_np.py, itself pinned by finite differences in tests/test_backward_oracle.py). The oracle
re-runs the blend loop in float64 on the GPU's own forward state (means2D, conics, colours, sorted lists)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 2e-4       # float32 accumulation in atomics (order varies) against a float64 sum


def _close(got, exp, what, rtol=RTOL):
    scale = max(1e-6, float(np.abs(exp).max()))
    err = float(np.abs(got - exp).max())
    assert err <= rtol * scale, f"{what}: max abs err {err} at scale {scale}"


def _forward_state(r):
    g = {k: v.cpu().numpy() for k, v in r.map_geometry_state().items()}
    im = {k: v.cpu().numpy() for k, v in r.map_image_state().items()}
    b = {k: v.cpu().numpy() for k, v in r.map_binning_state().items()}
    return g, im, b


@pytest.mark.parametrize("w,h,n,seed,bg,feed", [(64, 48, 300, 1, (0.0, 0.0, 0.0), "sorted"),
                                                (100, 70, 1200, 2, (0.2, 0.5, 0.9), "sorted"),
                                                (100, 70, 1200, 2, (0.2, 0.5, 0.9), "blocks"),
                                                (200, 136, 7000, 3, (0.1, 0.0, 0.3), "blocks"),
                                                (200, 136, 7000, 3, (0.1, 0.0, 0.3), "overlap")])
def test_backward_matches_float64_oracle(w, h, n, seed, bg, feed):
    """feed = "blocks": the forward call ran with GSR_FLAG_NO_SORTED_LISTS, so gsr_backward walks the tile lists out of
    the block lists (several units per block and several blocks in the last case); the oracle gets the sorted list of
    an ordinary call on the same inputs."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    from oracle import backward_np as B
    scene = scenes.garden_like_scene(n, seed=seed)
    scene["means3D"][:, :3] *= 0.25
    cam = camera.default_camera(w, h, near=0.05, far=50.0)
    r = SplatRasterizer(w, h, background=bg)
    r.configure_from_scene(scene)
    # ("overlap": GSR_FLAG_OVERLAP_EMIT — the blend ran on the library's second stream beside the emission; the backward
    # call, on the caller's stream, must find the image state and the sorted lists complete)
    img = r.draw(cam, plan="auto" if feed == "sorted" else "blocks", overlap_emit=feed == "overlap").cpu().numpy()
    assert r.last_num_rendered > 0
    g, im, b = _forward_state(r)
    if feed == "blocks":
        r.map_binning_state()["values"].fill_(-1)
        assert np.array_equal(r.draw(cam, plan="blocks", sorted_lists=False).cpu().numpy(), img)
        assert r.last_plan == "blocks" and not r.last_lists_written
        assert bool((r.map_binning_state()["values"] == -1).all())
    rng = np.random.default_rng(seed)
    dL = rng.normal(size=(3, h, w)).astype(np.float32)
    got = {k: v.cpu().numpy() for k, v in r.backward(torch.from_numpy(dL)).items()}

    ranges = im["ranges"].view(np.uint32).astype(np.int64)
    plist = b["values"].view(np.uint32).astype(np.int64)
    out64, ft64, nc64 = B.blend_forward(g["means2D"], g["conicOpacity"], g["rgb"], ranges, plist, w, h, bg)
    assert np.abs(out64 - img).max() <= 1e-4
    # the contributing set must be the same one (threshold flips between exp implementations would show here)
    assert (nc64 != im["nContrib"].view(np.uint32)).sum() <= 2
    exp = B.blend_backward(g["means2D"], g["conicOpacity"], g["rgb"], ranges, plist, nc64, ft64, w, h, bg, dL)
    _close(got["dL_dmean2D"], exp["dL_dmean2D"], "dL_dmean2D")
    _close(got["dL_dconic_opacity"][:, :3], exp["dL_dconic"], "dL_dconic")
    _close(got["dL_dconic_opacity"][:, 3], exp["dL_dopacity"], "dL_dopacity")
    _close(got["dL_dcolors"], exp["dL_dcolor"], "dL_dcolors")
    assert np.abs(exp["dL_dmean2D"]).max() > 0 and np.abs(exp["dL_dconic"]).max() > 0

    # covariance / colour chain down to the inputs, Gaussian by Gaussian, fed with the GPU's own upstream gradients
    from helpers import check_backward_chain
    vis = np.nonzero(g["radii"] > 0)[0]
    mags = check_backward_chain(got, g, scene, cam, w, h, vis)
    assert all(m > 0 for m in mags)
    culled = g["radii"] <= 0
    for k in ("dL_dcov3D", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
        assert (got[k][culled] == 0).all(), k
    assert (got["dL_dmeans3D"][:, 3] == 0).all() and (got["dL_dscales"][:, 3] == 0).all()
    sh = got["dL_dshs"]
    assert np.allclose(sh[:, :3], 0.4 * got["dL_dcolors"] * (g["radii"] > 0)[:, None], rtol=1e-6, atol=0)
    assert (sh[:, 3:] == 0).all()


@pytest.mark.parametrize("w,h,n,seed,deg", [(96, 64, 600, 4, 3), (64, 48, 300, 6, 1), (80, 80, 500, 8, 0)])
def test_backward_of_the_upstream_profile_matches_float64_oracle(w, h, n, seed, deg):
    """GSR_FLAG_SEMANTICS_INRIA through gsr_backward: the render backward on the forward state of that profile
    (transmittance cut-off 1e-4), then its own chain — SH colour with the view direction and the clamp, raw quaternion,
    two focal lengths — against oracle/backward_np.py (inria_*, pinned by finite differences on the CPU)."""
    import torch
    from helpers import check_backward_chain_inria
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    from oracle import backward_np as B
    scene = scenes.garden_like_scene(n, seed=seed)
    scene["means3D"][:, :3] *= 0.25
    rng = np.random.default_rng(seed)
    scene["shs"] = rng.normal(0, 0.35, (n, 48)).astype(np.float32)                 # [N][16][3]
    scene["rotations"] *= rng.uniform(0.7, 1.4, (n, 1)).astype(np.float32)         # not unit: the profile does not normalise
    cam = camera.default_camera(w, h, near=0.05, far=50.0)
    bg = (0.2, 0.5, 0.9)
    r = SplatRasterizer(w, h, background=bg)
    r.configure_from_scene(scene)
    img = r.draw(cam, semantics="inria", sh_degree=deg).cpu().numpy()
    assert r.last_num_rendered > 0
    g, im, b = _forward_state(r)
    st = r.map_geometry_state()
    dL = rng.normal(size=(3, h, w)).astype(np.float32)
    got = {k: v.cpu().numpy() for k, v in r.backward(torch.from_numpy(dL), semantics="inria", sh_degree=deg).items()}
    ranges = im["ranges"].view(np.uint32).astype(np.int64)
    plist = b["values"].view(np.uint32).astype(np.int64)
    out64, ft64, nc64 = B.blend_forward(g["means2D"], g["conicOpacity"], g["rgb"], ranges, plist, w, h, bg, t_cutoff=1e-4)
    assert np.abs(out64 - img).max() <= 1e-4
    assert (nc64 != im["nContrib"].view(np.uint32)).sum() <= 2
    exp = B.blend_backward(g["means2D"], g["conicOpacity"], g["rgb"], ranges, plist, nc64, ft64, w, h, bg, dL)
    _close(got["dL_dmean2D"], exp["dL_dmean2D"], "dL_dmean2D")
    _close(got["dL_dconic_opacity"][:, :3], exp["dL_dconic"], "dL_dconic")
    _close(got["dL_dconic_opacity"][:, 3], exp["dL_dopacity"], "dL_dopacity")
    _close(got["dL_dcolors"], exp["dL_dcolor"], "dL_dcolors")
    # the chain
    import ctypes as C
    from gsrast_amd import _capi
    gst = _capi.GeometryState()
    r.lib.gsr_geometry_from_chunk(r.geom.base(), n, C.byref(gst))
    clamped = r.geom.view(gst.clamped, 3 * n, torch.uint8).view(n, 3).cpu().numpy().astype(bool)
    vis = np.nonzero(g["radii"] > 0)[0]
    clamped[g["radii"] <= 0] = False
    mags = check_backward_chain_inria(got, g, scene, cam, w, h, vis, deg, clamped)
    assert all(m > 0 for m in mags)
    culled = g["radii"] <= 0
    for k in ("dL_dcov3D", "dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dshs"):
        assert (got[k][culled] == 0).all(), k
    assert (got["dL_dshs"][:, 3 * (deg + 1) ** 2:] == 0).all()                     # nothing beyond the degree in use
    if deg == 3:
        assert clamped[vis].any()                                                   # the clamp is exercised


def test_backward_of_a_single_gaussian_closed_form():
    """One isotropic Gaussian on the optical axis, dL_dout = 1 on one channel: the colour gradient is the sum of
    alpha over the pixels, the opacity gradient the sum of G * c (no transmittance in front, nothing behind)."""
    import torch
    from helpers import single_gaussian_scene
    from gsrast_amd import camera
    from gsrast_amd.rasterizer import SplatRasterizer
    scene = single_gaussian_scene(pos=(0.0, 0.0, 0.0), scale=0.2, opacity=0.6, dc=(1.0, 0.0, 0.0), n=2)
    scene["means3D"][1, :3] = (0.0, 0.0, -50.0)          # second one behind the camera: culled, zero gradients
    cam = camera.default_camera(64, 64)
    r = SplatRasterizer(64, 64)
    r.configure_from_scene(scene)
    r.draw(cam)
    st = r.map_geometry_state()
    m, co, col = st["means2D"][0].cpu().numpy(), st["conicOpacity"][0].cpu().numpy(), st["rgb"][0].cpu().numpy()
    dL = torch.zeros((3, 64, 64)); dL[0] = 1.0
    got = {k: v.cpu().numpy() for k, v in r.backward(dL).items()}
    ys, xs = np.mgrid[0:64, 0:64].astype(np.float64)
    dx, dy = m[0] - xs, m[1] - ys
    power = -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
    G = np.exp(power)
    alpha = np.minimum(0.99, co[3] * G)
    live = (power <= 0) & (alpha >= 1.0 / 255.0)
    assert abs(got["dL_dcolors"][0, 0] - (alpha * live).sum()) <= 1e-4 * (alpha * live).sum()
    exp_op = (G * col[0] * live).sum()
    assert abs(got["dL_dconic_opacity"][0, 3] - exp_op) <= 1e-4 * exp_op
    assert abs(got["dL_dmean2D"][0]).max() <= 1e-3 * exp_op     # symmetric footprint around an integer-aligned centre
    for k in ("dL_dmean2D", "dL_dconic_opacity", "dL_dcolors", "dL_dcov3D"):
        assert (got[k][1] == 0).all()


def test_backward_rejects_bad_arguments():
    import ctypes as C
    from gsrast_amd import _capi
    L = _capi.lib()
    a = _capi.BackwardArgs()
    a.struct_size = C.sizeof(_capi.BackwardArgs) - 4
    assert L.gsr_backward(C.byref(a)) == _capi.GSR_ERR_INVALID_ARG
    a.struct_size = C.sizeof(_capi.BackwardArgs)
    assert L.gsr_backward(C.byref(a)) == _capi.GSR_ERR_INVALID_ARG and L.gsr_last_error() == _capi.GSR_ERR_INVALID_ARG


def _base_sums(r, dl):
    import torch
    out = r.backward(dl, with_cov3D=False)
    return {k: out[k].clone() for k in ("dL_dmean2D", "dL_dconic_opacity", "dL_dcolors")}


def _assert_same_sums(got, ref, noise, what):
    for k, v in ref.items():
        scale = float(v.abs().max()) + 1e-30
        err = float((got[k] - v).abs().max())
        assert err <= max(4.0 * noise[k], 2e-4 * scale), (what, k, err, noise[k], scale)
        assert float(v.abs().sum()) > 0, (what, k)


def test_backward_from_the_block_lists_corner_cases():
    """The block-list feed of gsr_backward (csrc/blockbin.hpp BlockFeed) where its bookkeeping could go wrong: a second call
    on the same forward state (per-entry sums cleared again), a call confined to a band of tile rows, a forward state that
    a later sort-plan call replaced, and a frame of tiny splats under a forced block plan, where the per-entry sums do not
    fit (E > R / 4.5) and every tile falls back to direct atomics. Reference: the sorted-list path after a sort-plan call."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    W, H = 640, 400
    dl = torch.from_numpy(np.random.default_rng(11).normal(size=(3, H, W)).astype(np.float32))
    for name, scene, cam in (
            ("garden", scenes.garden_like_scene(120_000, seed=43), camera.default_camera(W, H, near=0.05, far=80.0)),
            ("tiny splats", scenes.stress_scene(400_000, seed=44), camera.default_camera(W, H, near=0.1, far=100.0, position=(0.0, 0.0, -25.0)))):
        r = SplatRasterizer(W, H, background=(0.3, 0.1, 0.2))
        r.configure_from_scene(scene)
        img = r.draw(cam, plan="sort").clone()
        assert r.last_plan == "sort" and r.last_num_rendered > 0
        ref = _base_sums(r, dl)
        again = _base_sums(r, dl)
        noise = {k: float((again[k] - v).abs().max()) for k, v in ref.items()}
        # block plan, sorted lists written: shallow blocks take the block lists, twice in a row
        assert torch.equal(r.draw(cam, plan="blocks"), img) and r.last_plan == "blocks"
        _assert_same_sums(_base_sums(r, dl), ref, noise, name + ": block plan")
        _assert_same_sums(_base_sums(r, dl), ref, noise, name + ": block plan, second call")
        # without the sorted lists
        r.draw(cam, plan="blocks", sorted_lists=False)
        assert not r.last_lists_written
        _assert_same_sums(_base_sums(r, dl), ref, noise, name + ": no sorted lists")
        # a sort-plan call in between: the block lists of the earlier call must not be used any more
        r.draw(cam, plan="blocks", sorted_lists=False)
        r.draw(cam, plan="sort")
        _assert_same_sums(_base_sums(r, dl), ref, noise, name + ": sort plan after a block-plan call")
    # a band of tile rows (the sharded call): forward and backward confined to rows [7, 19) of 25
    r.draw(cam, plan="sort", tile_rows=(7, 19))
    out = r.backward(dl, with_cov3D=False, tile_rows=(7, 19))
    ref = {k: out[k].clone() for k in ("dL_dmean2D", "dL_dconic_opacity", "dL_dcolors")}
    out = r.backward(dl, with_cov3D=False, tile_rows=(7, 19))
    noise = {k: float((out[k] - v).abs().max()) for k, v in ref.items()}
    for lists in (True, False):
        r.draw(cam, plan="blocks", tile_rows=(7, 19), sorted_lists=lists)
        out = r.backward(dl, with_cov3D=False, tile_rows=(7, 19))
        _assert_same_sums(out, ref, noise, f"band, sorted lists {lists}")


def _receipt_scene():
    from gsrast_amd import camera, scenes
    W, H = 400, 272
    return (W, H, scenes.garden_like_scene(60_000, seed=47), camera.default_camera(W, H, near=0.05, far=80.0),
            camera.default_camera(W, H, near=0.05, far=80.0, position=(0.5, 0.2, -7.0)))


_RECEIPT_MODES = {"sorted": dict(plan="sort"), "blocks": dict(plan="blocks"),
                  "blocks, no sorted lists": dict(plan="blocks", sorted_lists=False)}


@pytest.mark.parametrize("mode", list(_RECEIPT_MODES))
def test_backward_follows_the_receipt_not_the_last_call(mode):
    """What gsr_backward reads is decided by the gsr_forward_receipt it is handed and by the caller-owned chunks, not by
    anything the library remembers of its last call (round 2: thread-local last-call state; a backward issued from another
    host thread — what torch.autograd does — or after a second rasterizer's forward walked unwritten lists and returned
    GSR_OK). Here: forward on thread A / backward on thread B; forward A, forward B, backward A; each must give the sums of
    the plain same-thread sequence."""
    import threading
    import torch
    from gsrast_amd.rasterizer import SplatRasterizer
    W, H, scene, cam_a, cam_b = _receipt_scene()
    kw = _RECEIPT_MODES[mode]
    dl = torch.from_numpy(np.random.default_rng(5).normal(size=(3, H, W)).astype(np.float32)).cuda()
    ra, rb = SplatRasterizer(W, H, background=(0.1, 0.2, 0.3)), SplatRasterizer(W, H, background=(0.1, 0.2, 0.3))
    ra.configure_from_scene(scene)
    rb.configure_from_scene(scene)
    # reference: forward + backward of A, nothing in between, one thread
    ra.draw(cam_a, **kw)
    assert ra.last_num_rendered > 0 and ra.last_lists_written == kw.get("sorted_lists", True)
    ref = _base_sums(ra, dl)
    noise = {k: float((_base_sums(ra, dl)[k] - v).abs().max()) for k, v in ref.items()}

    # (1) forward here, backward on another host thread (its own thread-local state in the library: none of A's)
    ra.draw(cam_a, **kw)
    box = {}

    def worker():
        try:
            box["sums"] = _base_sums(ra, dl)
            ra.poll_async_error()                      # any thread may ask about a call
        except Exception as e:                         # noqa: BLE001
            box["error"] = e
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert "error" not in box, box.get("error")
    _assert_same_sums(box["sums"], ref, noise, mode + ": backward on another thread")

    # (2) a second rasterizer's forward (other camera, other plan) between A's forward and A's backward
    ra.draw(cam_a, **kw)
    receipt_a = ra.last_receipt
    rb.draw(cam_b, plan="blocks" if kw["plan"] == "sort" else "sort")
    rb.draw(cam_b, plan="blocks", sorted_lists=False)
    out = ra.backward(dl, with_cov3D=False, receipt=receipt_a)
    _assert_same_sums(out, ref, noise, mode + ": another rasterizer's forward in between")
    # ... and B's own backward is B's
    rb.draw(cam_b, plan="sort")
    ref_b = _base_sums(rb, dl)
    rb.draw(cam_b, **kw)
    ra.draw(cam_a, plan="sort")
    out_b = rb.backward(dl, with_cov3D=False)
    noise_b = {k: 1e-30 for k in ref_b}
    _assert_same_sums(out_b, ref_b, noise_b, mode + ": B after A's forward")


def test_backward_without_a_fitting_receipt_is_refused_never_garbage():
    import ctypes as C
    import torch
    from gsrast_amd import _capi
    from gsrast_amd.rasterizer import SplatRasterizer
    W, H, scene, cam_a, cam_b = _receipt_scene()
    dl = torch.from_numpy(np.random.default_rng(6).normal(size=(3, H, W)).astype(np.float32)).cuda()
    ra, rb = SplatRasterizer(W, H), SplatRasterizer(W, H)
    ra.configure_from_scene(scene)
    rb.configure_from_scene(scene)
    ra.draw(cam_a, plan="sort")
    ref = _base_sums(ra, dl)
    noise = {k: float((_base_sums(ra, dl)[k] - v).abs().max()) for k, v in ref.items()}
    # no receipt, sorted lists in place (the reference's contract): served, from the sorted lists
    for plan in ("sort", "blocks"):
        ra.draw(cam_a, plan=plan)
        _assert_same_sums(ra.backward(dl, with_cov3D=False, receipt=False), ref, noise, f"no receipt after a {plan}-plan call")
    # no receipt, lists skipped: the stamp in values[0] says so
    ra.draw(cam_a, plan="blocks", sorted_lists=False)
    assert int(ra.map_binning_state()["values"][0].item()) & 0xFFFFFFFF == _capi.GSR_LISTS_SKIPPED_STAMP
    with pytest.raises(_capi.GsrError) as e:
        ra.backward(dl, with_cov3D=False, receipt=False)
    assert e.value.code == _capi.GSR_ERR_INVALID_ARG
    # ... and with the receipt the same state is served
    _assert_same_sums(ra.backward(dl, with_cov3D=False), ref, noise, "receipt after a call without sorted lists")
    # a receipt of another rasterizer's call (other chunks), of another size, of another band, a forged magic
    rb.draw(cam_b, plan="blocks", sorted_lists=False)
    with pytest.raises(_capi.GsrError) as e:
        ra.backward(dl, with_cov3D=False, receipt=rb.last_receipt)
    assert e.value.code == _capi.GSR_ERR_INVALID_ARG
    for field, value in (("width", W + 16), ("num_gaussians", 5), ("tile_row_end", 3), ("magic", 0x12345678)):
        bad = ra.last_receipt.copy()
        setattr(bad, field, value)
        with pytest.raises(_capi.GsrError) as e:
            ra.backward(dl, with_cov3D=False, receipt=bad)
        assert e.value.code == _capi.GSR_ERR_INVALID_ARG, field
    # polling: NULL and a receipt without the magic are refused; a receipt whose slot has been handed on says so
    L = _capi.lib()
    assert L.gsr_poll_async_error(None) == _capi.GSR_ERR_INVALID_ARG
    assert L.gsr_poll_async_error(C.byref(_capi.ForwardReceipt())) == _capi.GSR_ERR_INVALID_ARG
    old = ra.last_receipt.copy()
    for _ in range(70):                               # more calls than there are error-word slots
        ra.draw(cam_a, plan="blocks", sync=False)
    torch.cuda.synchronize()
    assert L.gsr_poll_async_error(C.byref(old)) == _capi.GSR_ERR_STALE_RECEIPT    # (64 slots: nothing is known about that call any more)
    assert L.gsr_poll_async_error(C.byref(ra.last_receipt)) == _capi.GSR_OK
    assert ra.last_receipt.serial >= old.serial + 70       # (the serial counts every call of this thread on this device)


def test_double_sums_are_reproducible_and_agree_with_the_float_sums():
    """gsr_backward_args.sums_f64: the twelve per-Gaussian sums accumulated in double and rounded once. Two runs give the same
    bits (float atomics arrive in another order every run: their sums differ in the last places), with and without the chain
    behind them; the scratch is left zero; and the float mode agrees to the accuracy float sums have."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    w, h = 640, 360
    scene = scenes.garden_like_scene(60_000, seed=31)
    scene["means3D"][:, :3] *= 0.3
    scene["scales"][:200, :3] *= 30.0                            # a few splats that fill the frame: thousands of terms per sum
    r = SplatRasterizer(w, h, background=(0.1, 0.2, 0.3))
    r.configure_from_scene(scene)
    r.draw(camera.default_camera(w, h, near=0.05, far=50.0))
    dL = torch.randn((3, h, w), generator=torch.Generator().manual_seed(5)).cuda()
    a = {k: v.clone() for k, v in r.backward(dL, wide_sums=True).items()}
    assert float(r._sums_f64.abs().max()) == 0.0                 # left zero for the next call
    b = {k: v.clone() for k, v in r.backward(dL, wide_sums=True).items()}
    for k in a:
        assert torch.equal(a[k], b[k]), k
    base = {k: v.clone() for k, v in r.backward(dL, wide_sums=True, with_cov3D=False).items()}
    assert float(r._sums_f64.abs().max()) == 0.0
    for k in ("dL_dmean2D", "dL_dconic_opacity", "dL_dcolors"):
        assert torch.equal(base[k], a[k]), k                     # the same sums whether or not the chain runs behind them
    f = r.backward(dL, wide_sums=False)
    for k in a:
        scale = float(a[k].abs().max())
        assert scale > 0 and float((f[k] - a[k]).abs().max()) <= 2e-4 * scale, (k, float((f[k] - a[k]).abs().max()), scale)
    culled = r.map_geometry_state()["radii"] <= 0
    for k in a:
        assert bool((a[k][culled] == 0).all()), k


def test_optional_outputs_give_the_same_numbers_and_leave_the_others_alone():
    """With sums_f64 every output of gsr_backward is optional (the chain is bound by what it writes): a subset gives the bits
    the full call gives for its members, the arrays not asked for are not written, the scratch is left zero — for the set
    BASELINE config 5 names, for the gradients of the inputs alone, and for one array on its own; and the same subset
    without sums_f64 is refused where the float arrays of the sums would be missing."""
    import ctypes as C
    import torch
    from gsrast_amd import _capi, camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    w, h = 640, 360
    scene = scenes.garden_like_scene(50_000, seed=33)
    scene["means3D"][:, :3] *= 0.3
    r = SplatRasterizer(w, h, background=(0.1, 0.2, 0.3))
    r.configure_from_scene(scene)
    r.draw(camera.default_camera(w, h, near=0.05, far=50.0))
    dL = torch.randn((3, h, w), generator=torch.Generator().manual_seed(6)).cuda()
    full = {k: v.clone() for k, v in r.backward(dL).items()}
    assert all(float(v.abs().max()) > 0 for v in full.values())
    cache = r._bw_out_by_semantics["gscuda"]
    for subset in (("dL_dmean2D", "dL_dcov3D", "dL_dshs"),
                   ("dL_dmean2D", "dL_dconic_opacity", "dL_dshs", "dL_dmeans3D", "dL_dscales", "dL_drotations"),
                   ("dL_dmeans3D",), ("dL_dcolors",), ("dL_dshs",)):
        for k, v in cache.items():
            if k != "dL_dshs":
                v.fill_(7.0)                                   # (dL_dshs: floats 16..47 are the caller's zeros)
        got = r.backward(dL, outputs=subset)
        assert set(got) == set(subset)
        for k in subset:
            assert torch.equal(got[k], full[k]), (subset, k)
        for k, v in cache.items():
            if k not in subset and k != "dL_dshs":
                assert bool((v == 7.0).all()), (subset, k)     # not asked for: not written
        assert float(r._sums_f64.abs().max()) == 0.0
    with pytest.raises(AssertionError):
        r.backward(dL, outputs=("dL_dcov3D",), wide_sums=False)
    # the C ABI itself: without sums_f64 the three arrays of the sums are required
    a = _capi.BackwardArgs()
    a.struct_size = C.sizeof(_capi.BackwardArgs)
    a.num_gaussians, a.width, a.height = r.num_gaussians, w, h
    for name in ("background", "means2D", "conic_opacity", "colors", "ranges", "n_contrib", "final_t", "dL_dout_color", "point_list"):
        setattr(a, name, dL.data_ptr())                        # (refused before anything is read)
    a.dL_dcov3D = cache["dL_dcov3D"].data_ptr()
    assert r.lib.gsr_backward(C.byref(a)) == _capi.GSR_ERR_INVALID_ARG


def test_backward_of_a_frame_without_instances_is_all_zero():
    """R == 0: the forward call leaves even the tile ranges unwritten (GSCuda.cu:775-778); with its receipt gsr_backward
    returns zero gradients instead of walking whatever an earlier frame left in the chunks."""
    import torch
    from gsrast_amd import camera
    from gsrast_amd.rasterizer import SplatRasterizer
    from helpers import single_gaussian_scene
    W, H = 128, 96
    r = SplatRasterizer(W, H)
    r.configure_from_scene(single_gaussian_scene(n=40, scale=0.3))
    cam = camera.default_camera(W, H, near=0.05, far=50.0)
    r.draw(cam)
    assert r.last_num_rendered > 0
    dl = torch.ones((3, H, W), device="cuda")
    assert float(r.backward(dl)["dL_dcolors"].abs().sum()) > 0
    away = camera.default_camera(W, H, near=0.05, far=50.0, position=(0.0, 0.0, 5.0))   # the scene is behind this camera
    r.draw(away)
    assert r.last_num_rendered == 0 and r.last_receipt.magic != 0 and not r.last_receipt.binning_chunk
    out = r.backward(dl)
    for k, v in out.items():
        assert float(v.abs().sum()) == 0.0, k
