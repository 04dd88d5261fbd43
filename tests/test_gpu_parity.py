"""GPU parity: the HIP path through the C ABI against the CPU oracle and the golden fixture.

Integer / index outputs must be bit-exact. The preprocess stage keeps the reference's
float32 operation order, so its float outputs are compared exactly too. So does the blend
stage, exponential included (glibc's expf restated: blend_core.hpp): transmittance and
nContrib are compared bit for bit, pixels at 2e-6 (the colour sums are fused; the north
star's tolerance is 1e-4 abs per channel).
"""
import numpy as np
import pytest

from helpers import load_golden, image_report, single_gaussian_scene, assert_blend_parity

pytestmark = pytest.mark.gpu

TOL = 1e-4      # BASELINE.json north_star: "within 1e-4 abs per channel"


def _gpu():
    import torch
    from gsrast_amd.rasterizer import SplatRasterizer
    assert torch.cuda.is_available(), "no HIP device"
    return torch, SplatRasterizer


PLAN = "auto"


@pytest.fixture(autouse=True, params=["sort", "blocks"])
def binning_plan(request):
    """Every frame-level test runs under both binning plans (GSR_FLAG_PLAN_SORT / _BLOCKS): they must
    give the same sorted keys / values / ranges / pixels."""
    global PLAN
    PLAN = request.param
    yield
    PLAN = "auto"


def _run(scene, cam, bg=(0.0, 0.0, 0.0), use_rects=True, **kw):
    torch, SplatRasterizer = _gpu()
    kw.setdefault("plan", PLAN)
    r = SplatRasterizer(cam.width, cam.height, background=bg)
    r.configure_from_scene(scene, use_rects=use_rects)
    # zero the chunks once so fields the reference leaves unwritten compare equal to the oracle's zeros
    r.draw(cam, **kw)
    for cb in (r.geom, r.image, r.binning):
        if cb.tensor is not None:
            cb.tensor.zero_()
    if r.rects is not None:
        r.rects.zero_()
    r.out_color.zero_()
    img = r.draw(cam, count_staged=True, **kw).cpu().numpy().copy()
    return r, img


def _compare_all(r, img, exp, n, expect_plan=None, bitwise_t=True):
    g = {k: v.cpu().numpy() for k, v in r.map_geometry_state().items()}
    assert np.array_equal(g["radii"], exp["radii"])
    assert np.array_equal(g["tilesTouched"].view(np.uint32), exp["tilesTouched"])
    assert np.array_equal(g["pointOffsets"].view(np.uint32), exp["pointOffsets"])
    if exp.get("rects") is not None:
        assert np.array_equal(r.rects.cpu().numpy(), exp["rects"])
    for k in ("means2D", "depths", "cov3D", "rgb", "conicOpacity"):
        assert np.array_equal(g[k], exp[k]), f"{k}: max diff {np.abs(g[k] - exp[k]).max()}"
    assert r.last_num_rendered == exp["num_rendered"]
    if exp["num_rendered"] > 0:
        b = {k: v.cpu().numpy() for k, v in r.map_binning_state().items()}
        # The reference emits the pairs in index order. The HIP path sorts the depth half of the key
        # per Gaussian before duplication and writes the tile-column pass directly, so its
        # "unsorted" arrays hold the same multiset of pairs ordered by (tile column, depth) — or by
        # depth alone on grids wider than 255 tiles. (The block plan writes the sorted lists
        # directly and keeps its block lists in the two "unsorted" arrays: nothing to compare.)
        gx, gy = (r.width + 15) // 16, (r.height + 15) // 16
        assert r.last_plan == (expect_plan or ("generic" if (gx > 255 or gy > 255) else PLAN))
        ku, vu = b["keys_unsorted"].view(np.uint64), b["values_unsorted"].view(np.uint32)
        if r.last_plan != "blocks":
            o_g, o_e = np.lexsort((vu, ku)), np.lexsort((exp["values_unsorted"], exp["keys_unsorted"]))
            assert np.array_equal(ku[o_g], exp["keys_unsorted"][o_e])
            assert np.array_equal(vu[o_g], exp["values_unsorted"][o_e])
        depth_half = (ku & np.uint64(0xFFFFFFFF)).astype(np.int64)
        if r.last_plan == "blocks":
            pass
        elif gx <= 255 and gy <= 255:
            col = ((ku >> np.uint64(32)) % np.uint64(gx)).astype(np.int64)
            assert bool((np.diff(col) >= 0).all())
            same_col = np.diff(col) == 0
            assert bool((np.diff(depth_half)[same_col] >= 0).all())
        else:
            assert bool((np.diff(depth_half) >= 0).all())
        assert np.array_equal(b["keys"].view(np.uint64), exp["keys"])
        assert np.array_equal(b["values"].view(np.uint32), exp["values"])
    im = {k: v.cpu().numpy() for k, v in r.map_image_state().items()}
    assert np.array_equal(im["ranges"].view(np.uint32), exp["ranges"])
    assert_blend_parity(img, im["finalT"], im["nContrib"], exp, f"{r.width}x{r.height} N={n} plan={r.last_plan}", bitwise_t)
    assert r.last_records_staged == exp["records_staged"]


def test_config1_against_golden_fixture():
    scene, cam, bg, exp = load_golden()
    r, img = _run(scene, cam, bg)
    _compare_all(r, img, exp, scene["means3D"].shape[0], bitwise_t=False)     # (the fixture is the numpy restatement's: its exp is not libm's)
    from oracle import cpu_oracle
    _compare_all(r, img, cpu_oracle.forward(scene, cam, bg), scene["means3D"].shape[0])
    # allocator contract: geometry, image, binning; once each, in that order (GSCuda.cu:723-784)
    L = r.lib
    n, P, R = r.num_gaussians, cam.width * cam.height, r.last_num_rendered
    assert r.geom.calls[-1] == L.gsr_required_geometry(n)
    assert r.image.calls[-1] == L.gsr_required_image(P) + 128
    assert r.binning.calls[-1] == L.gsr_required_binning(R) + 128
    assert len(r.geom.calls) == len(r.image.calls) == len(r.binning.calls) == 2


@pytest.mark.parametrize("w,h,n,seed", [(200, 120, 3000, 7), (333, 257, 20000, 11), (64, 48, 500, 3)])
def test_anisotropic_scenes_against_oracle(w, h, n, seed):
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    scene = scenes.garden_like_scene(n, seed=seed)
    scene["means3D"][:, :3] *= 0.25
    cam = camera.default_camera(w, h, near=0.05, far=50.0)
    bg = (0.1, 0.2, 0.3)
    exp = cpu_oracle.forward(scene, cam, bg)
    r, img = _run(scene, cam, bg)
    _compare_all(r, img, exp, n)


@pytest.mark.parametrize("eye", [(0.5, -0.2, -4.0), (0.0, 0.0, -48.0)])
def test_trained_like_scene_against_oracle(eye):
    """Flat splats on surfaces, a heavy-tailed scale distribution, bimodal opacity and 500 huge background splats
    (scenes.trained_like): from inside, and from 48 units away, where a Gaussian touches three tiles on average and the
    background splats a thousand each."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    scene = scenes.trained_like(60_000, seed=46)
    cam = camera.default_camera(640, 368, near=0.02, far=120.0, position=eye)
    bg = (0.1, 0.2, 0.3)
    exp = cpu_oracle.forward(scene, cam, bg, threads=8)
    assert exp["num_rendered"] > 50_000
    r, img = _run(scene, cam, bg)
    _compare_all(r, img, exp, 60_000)


@pytest.mark.parametrize("scale,big_every", [(0.008, 0), (0.012, 0), (0.012, 700), (0.015, 0), (0.035, 0)])
def test_chunks_around_the_sort_plans_placement_limits(scale, big_every):
    """emit_chunk_kernel (sort plan) writes a chunk of 512 depth-consecutive Gaussians one of two ways: up to 2048 keys, none
    of its Gaussians with more than 32, every Gaussian places its own keys; otherwise every key looks for its Gaussian. Isotropic
    splats of a size that puts the chunks' key counts on either side of 2048 (a frame's chunks differ: near ones are
    heavier), with and without a splat of hundreds of tiles in some chunks — lists against the oracle bit for bit."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    n = 12_000
    scene = scenes.isotropic_scene(n, seed=51)
    scene["scales"][:, :3] = scale
    if big_every:
        scene["scales"][::big_every, :3] = 0.6
    cam = camera.default_camera(640, 368, near=0.05, far=50.0, position=(0.0, 0.0, -3.0))
    bg = (0.0, 0.1, 0.0)
    exp = cpu_oracle.forward(scene, cam, bg, threads=8)
    tt = exp["tilesTouched"]
    assert exp["num_rendered"] > 2 * int((tt > 0).sum())
    r, img = _run(scene, cam, bg)
    _compare_all(r, img, exp, n)


def test_a_few_huge_splats_send_a_frame_of_small_ones_to_the_block_plan():
    """The plan follows the instances per visible Gaussian — of the splats that are small: with an eighth of the frame's
    instances in splats of 256 tiles and more (the scan counts them) the block plan is taken whatever the average
    (the sort plan's emission walks a Gaussian's keys chunk by chunk: 1.53 against 1.08 ms on 1 M flat splats + 500 huge ones)."""
    import torch
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    scene = scenes.trained_like(200_000, seed=47)
    w, h = 1920, 1080
    r = SplatRasterizer(w, h)
    r.configure_from_scene(scene)
    cam = camera.default_camera(w, h, near=0.02, far=200.0, position=(0.0, 0.0, -48.0))
    img = r.draw(cam).clone()
    tt = r.map_geometry_state()["tilesTouched"].to(torch.int64)
    R, V, big = int(tt.sum()), int((tt != 0).sum()), int(tt[tt >= 256].sum())
    assert R == r.last_num_rendered and R < 6 * V and 8 * big >= R          # (small on average, an eighth and more in big splats)
    assert r.last_plan == "blocks"
    # forced, both plans give the same picture; without its big splats the same frame takes the sort plan
    assert torch.equal(r.draw(cam, plan="sort").view(torch.int32), img.view(torch.int32)) and r.last_plan == "sort"
    small_ones = (tt < 256).cpu().numpy()
    r2 = SplatRasterizer(w, h)
    r2.configure_from_scene({k: np.ascontiguousarray(v[small_ones]) for k, v in scene.items()})
    r2.draw(cam)
    assert r2.last_num_rendered == R - big and r2.last_plan == "sort"


def test_grid_wider_than_255_tiles_uses_generic_digit_passes():
    """4112 x 40 -> 257 x 3 tiles: depth-ordered emission + 8-bit digit passes over the tile bits."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    scene = scenes.garden_like_scene(4000, seed=5)
    scene["means3D"][:, :3] *= 0.2
    scene["means3D"][:, 0] *= 12.0
    cam = camera.default_camera(4112, 40, near=0.05, far=50.0)
    exp = cpu_oracle.forward(scene, cam, (0.0, 0.1, 0.2))
    assert exp["num_rendered"] > 1000
    r, img = _run(scene, cam, (0.0, 0.1, 0.2))
    _compare_all(r, img, exp, 4000)


@pytest.mark.parametrize("w,h,n,seed,xs", [(2500, 200, 6000, 13, 8.0), (3000, 330, 9000, 17, 9.5)])
def test_grids_of_129_to_192_tile_columns(w, h, n, seed, xs):
    """157 / 188 tile columns: the sort plan's column tables are laid out for 192 columns (its per-key column search
    starts at 128), the block plan has 20 / 24 block columns and partial blocks on both edges."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    scene = scenes.garden_like_scene(n, seed=seed)
    scene["means3D"][:, :3] *= 0.2
    scene["means3D"][:, 0] *= xs
    cam = camera.default_camera(w, h, near=0.05, far=50.0)
    exp = cpu_oracle.forward(scene, cam, (0.2, 0.1, 0.0))
    assert exp["num_rendered"] > 5000
    r, img = _run(scene, cam, (0.2, 0.1, 0.0))
    _compare_all(r, img, exp, n)


def test_more_than_512_tile_blocks_fall_back_to_the_sort_plan():
    """4000 x 2400 -> 250 x 150 tiles = 32 x 19 = 608 blocks of 8 x 8 tiles: more than the block plan's tables hold
    (kMaxBlocks = 512), so the sort plan runs whatever the flag says — with 250 tile columns in its 256-column tables."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    scene = scenes.garden_like_scene(12000, seed=23)
    scene["means3D"][:, :3] *= 0.25
    cam = camera.default_camera(4000, 2400, near=0.05, far=50.0)
    exp = cpu_oracle.forward(scene, cam, (0.0, 0.0, 0.0))
    assert exp["num_rendered"] > 100000
    r, img = _run(scene, cam, (0.0, 0.0, 0.0))
    _compare_all(r, img, exp, 12000, expect_plan="sort")


def test_single_tile_row_and_single_tile_grids():
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    for w, h in ((100, 16), (16, 16), (16, 90)):
        scene = scenes.isotropic_scene(300, seed=8)
        cam = camera.default_camera(w, h)
        exp = cpu_oracle.forward(scene, cam)
        r, img = _run(scene, cam)
        _compare_all(r, img, exp, 300)


def test_radius_rect_path():
    from oracle import cpu_oracle
    scene, cam, bg, _ = load_golden()
    exp = cpu_oracle.forward(scene, cam, bg, use_rects=False)
    r, img = _run(scene, cam, bg, use_rects=False)
    _compare_all(r, img, exp, scene["means3D"].shape[0])


def test_nothing_visible_leaves_out_color_untouched():
    """R == 0: early return, outColor keeps whatever it held (GSCuda.cu:775-778)."""
    torch, SplatRasterizer = _gpu()
    from gsrast_amd import camera
    scene = single_gaussian_scene(pos=(0.0, 0.0, -50.0), n=3)      # behind the camera
    cam = camera.default_camera(64, 64)
    r = SplatRasterizer(64, 64)
    r.configure_from_scene(scene)
    r.out_color.fill_(0.25)
    r.draw(cam)
    assert r.last_num_rendered == 0
    assert bool((r.out_color == 0.25).all())
    assert len(r.binning.calls) == 0           # the binning allocator is not called when R == 0


def test_single_instance_draws_nothing():
    """R == 1: identifyTileRanges never closes the only tile (GSCuda.cu:515-537)."""
    from gsrast_amd import camera
    from oracle import cpu_oracle
    scene = single_gaussian_scene(pos=(0.5178, -0.5178, 0.0), scale=0.001, n=1)   # lands mid-tile at pixel (40, 40)
    cam = camera.default_camera(64, 64)
    exp = cpu_oracle.forward(scene, cam, (0.2, 0.3, 0.4))
    assert exp["num_rendered"] == 1
    r, img = _run(scene, cam, (0.2, 0.3, 0.4))
    assert r.last_num_rendered == 1
    assert np.array_equal(img, exp["out_color"])
    assert np.allclose(img[0], 0.2) and np.allclose(img[2], 0.4)


def test_equal_keys_keep_ascending_index_order():
    """Coincident Gaussians give identical (tile, depth) keys; the sort must be stable."""
    from gsrast_amd import camera
    from oracle import cpu_oracle
    scene = single_gaussian_scene(pos=(0.1, -0.2, 0.3), scale=0.15, opacity=0.3, n=300)
    scene["shs"][:, 0] = np.linspace(-1, 1, 300)
    cam = camera.default_camera(96, 96)
    exp = cpu_oracle.forward(scene, cam)
    r, img = _run(scene, cam)
    _compare_all(r, img, exp, 300)
    v = r.map_binning_state()["values"].cpu().numpy()
    k = r.map_binning_state()["keys"].cpu().numpy()
    same = k[1:] == k[:-1]
    assert same.any() and bool((v[1:][same] > v[:-1][same]).all())


def test_opaque_stack_terminates_early():
    """An opaque stack drives T below 0.001: done pixels, nContrib and R_f < R."""
    from gsrast_amd import camera
    from oracle import cpu_oracle
    n = 1200
    scene = single_gaussian_scene(pos=(0.0, 0.0, 0.0), scale=0.6, opacity=0.95, n=n)
    scene["means3D"][:, 2] = np.linspace(-1.0, 1.0, n)
    cam = camera.default_camera(64, 64)
    exp = cpu_oracle.forward(scene, cam)
    assert exp["records_staged"] < exp["num_rendered"]
    r, img = _run(scene, cam)
    _compare_all(r, img, exp, n)


def test_scan_matches_numpy_cumsum():
    torch, _ = _gpu()
    from gsrast_amd.rasterizer import inclusive_scan_u32
    rng = np.random.default_rng(5)
    for n in (1, 63, 4096, 4097, 1_000_003, 5_834_784):
        x = rng.integers(0, 50, size=n, dtype=np.int64).astype(np.uint32)
        got = inclusive_scan_u32(torch.from_numpy(x.view(np.int32)).cuda()).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, np.cumsum(x.astype(np.uint64)).astype(np.uint32)), n


def test_sort_matches_stable_argsort():
    torch, _ = _gpu()
    from gsrast_amd.rasterizer import sort_pairs
    rng = np.random.default_rng(9)
    for n, bits in ((1, 64), (255, 40), (4096, 45), (4097, 39), (1_234_567, 45), (3_000_001, 47)):
        tile = rng.integers(0, 1 << (bits - 32), size=n, dtype=np.uint64)
        depth = rng.integers(0, 1 << 12, size=n, dtype=np.uint64) << np.uint64(17)   # many ties
        keys = (tile << np.uint64(32)) | depth
        vals = np.arange(n, dtype=np.uint32)
        ko, vo = sort_pairs(torch.from_numpy(keys.view(np.int64)).cuda(),
                            torch.from_numpy(vals.view(np.int32)).cuda(), end_bit=bits)
        order = np.argsort(keys, kind="stable")
        if n > 1000:    # only the tile half [32, bits): depth order of equal tiles must survive
            k2, v2 = sort_pairs(torch.from_numpy(keys.view(np.int64)).cuda(),
                                torch.from_numpy(vals.view(np.int32)).cuda(), end_bit=bits, begin_bit=32)
            o2 = np.argsort(keys >> np.uint64(32), kind="stable")
            assert np.array_equal(k2.cpu().numpy().view(np.uint64), keys[o2]), (n, bits, "tile half")
            assert np.array_equal(v2.cpu().numpy().view(np.uint32), vals[o2]), (n, bits, "tile half")
        assert np.array_equal(ko.cpu().numpy().view(np.uint64), keys[order]), (n, bits)
        assert np.array_equal(vo.cpu().numpy().view(np.uint32), vals[order]), (n, bits)


def test_block_lists_longer_than_one_unit_with_ties_everywhere():
    """6000 coincident Gaussians over a 3 x 3-tile footprint straddling four 8 x 8-tile blocks: every block list
    has several 2048-entry units, every key of a tile is equal — the order inside a tile must still be the
    ascending Gaussian index, across unit and batch boundaries."""
    from gsrast_amd import camera
    from oracle import cpu_oracle
    n = 6000
    scene = single_gaussian_scene(pos=(0.02, -0.03, 0.0), scale=0.035, opacity=0.02, n=n)
    scene["shs"][:, 1] = np.linspace(-1, 1, n)
    cam = camera.default_camera(256, 256)          # 16 x 16 tiles = 2 x 2 blocks, the splat sits on their corner
    exp = cpu_oracle.forward(scene, cam)
    tiles = np.unique(exp["keys"] >> np.uint64(32))
    assert len(tiles) >= 4 and len({(int(t) % 16) // 8 + 2 * ((int(t) // 16) // 8) for t in tiles}) == 4
    r, img = _run(scene, cam)
    _compare_all(r, img, exp, n)
    v = r.map_binning_state()["values"].cpu().numpy()
    k = r.map_binning_state()["keys"].cpu().numpy()
    same = k[1:] == k[:-1]
    assert same.sum() >= n and bool((v[1:][same] > v[:-1][same]).all())


def test_one_splat_over_every_block_and_many_small_ones():
    """A Gaussian whose rectangle covers the whole 40 x 23-tile grid (15 blocks, partial ones at the right and
    bottom edge) among small ones; band-limited call included."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    scene = scenes.garden_like_scene(3000, seed=21)
    scene["means3D"][:, :3] *= 0.25
    scene["means3D"][0, :3] = (0.0, 0.0, -2.0)
    scene["scales"][0, :3] = 3.0
    cam = camera.default_camera(640, 360, near=0.05, far=50.0)
    exp = cpu_oracle.forward(scene, cam, (0.3, 0.3, 0.3))
    assert exp["tilesTouched"][0] == 40 * 23
    r, img = _run(scene, cam, (0.3, 0.3, 0.3))
    _compare_all(r, img, exp, 3000)
    full = img.copy()
    r.out_color.fill_(-1.0)
    for rows in ((0, 7), (7, 8), (8, 23)):
        r.draw(cam, tile_rows=rows, plan=PLAN)
    assert np.array_equal(r.out_color.cpu().numpy(), full)


@pytest.mark.parametrize("seed", range(12))
def test_random_small_frames_against_oracle(seed):
    """Random frame sizes (odd widths, partial tile blocks, one-row / one-column grids), splat counts and splat
    sizes: everything the oracle produces, bit for bit, under both binning plans (the module fixture)."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(17, 700)), int(rng.integers(17, 420))
    if seed % 4 == 0:
        h = int(rng.integers(1, 17))                 # a single tile row
    if seed % 4 == 1:
        w = int(rng.integers(1, 17))                 # a single tile column
    n = int(rng.integers(1, 4000))
    scene = scenes.garden_like_scene(n, seed=2000 + seed)
    scene["means3D"][:, :3] *= float(rng.uniform(0.1, 0.6))
    scene["scales"][:, :3] *= float(np.exp(rng.uniform(-1.0, 2.5)))      # from sub-pixel to screen-filling
    scene["opacities"][:] = rng.uniform(0.01, 1.0, n).astype(np.float32)
    cam = camera.default_camera(w, h, near=0.05, far=50.0)
    bg = tuple(float(v) for v in rng.uniform(0, 1, 3))
    exp = cpu_oracle.forward(scene, cam, bg)
    if exp["num_rendered"] == 0:
        pytest.skip("nothing visible for this seed")
    r, img = _run(scene, cam, bg)
    _compare_all(r, img, exp, n)


@pytest.mark.parametrize("seed", range(10))
def test_extreme_but_finite_inputs_against_oracle(seed):
    """Values a trained scene does not hold but a caller may pass: scales from 1e-8 to screen-filling and 1e6 : 1
    needles, quaternions far from unit length, opacities 0, 1, above 1 and below 0, splats on the camera plane
    (clip w = 0: 1 / (w + 0.001)), behind the eye and 1e5 units away, an eye inside the cloud looking anywhere. The
    integer outputs, every per-Gaussian float, both lists, the transmittance and nContrib must still be the oracle's bit for bit
    under both plans (the module fixture); pixels within 2e-6."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    rng = np.random.default_rng(7000 + seed)
    w, h = int(rng.integers(40, 500)), int(rng.integers(40, 300))
    n = int(rng.integers(200, 3000))
    scene = scenes.garden_like_scene(n, seed=7100 + seed)
    scene["means3D"][:, :3] *= 0.3
    pick = lambda frac: rng.random(n) < frac
    sc = scene["scales"]
    sc[pick(0.05), :3] = 1e-8
    sc[pick(0.05), :3] *= 1e3                                                # screen-filling
    needle = pick(0.1)
    sc[needle, 0] *= 1e3; sc[needle, 1] *= 1e-3                              # 1e6 : 1
    scene["rotations"][pick(0.1)] *= 1e3                                     # far from unit length (the preprocess re-normalises)
    scene["rotations"][pick(0.1)] *= 1e-3
    op = scene["opacities"]
    op[pick(0.05)] = 0.0; op[pick(0.05)] = 1.0; op[pick(0.03)] = 2.0; op[pick(0.03)] = -0.5; op[pick(0.05)] = 1e-9
    eye = np.array([0.0, 0.0, -1.5]) if seed % 2 else rng.uniform(-1.0, 1.0, 3)
    yaw, pitch = (0.0, 0.0) if seed % 2 else (float(rng.uniform(-3.1, 3.1)), float(rng.uniform(-1.0, 1.0)))
    front = np.array([np.cos(pitch) * np.sin(yaw), np.sin(pitch), np.cos(pitch) * np.cos(yaw)])
    m = scene["means3D"]
    on_plane = pick(0.03)                                                    # on the camera plane: forward distance exactly 0 (or as near as float32 goes)
    m[on_plane, :3] -= (((m[on_plane, :3] - eye) @ front)[:, None] * front[None, :]).astype(np.float32)
    m[pick(0.02), :3] *= 1e5                                                 # far outside the frustum's depth range
    cam = camera.first_person_camera(tuple(float(v) for v in eye), yaw, pitch, float(np.radians(45.0)), 0.01, 60.0, w, h, True)
    bg = tuple(float(v) for v in rng.uniform(0, 1, 3))
    exp = cpu_oracle.forward(scene, cam, bg)
    if exp["num_rendered"] == 0:
        pytest.skip("nothing visible for this seed")
    for k in ("means2D", "conicOpacity", "cov3D", "depths"):
        assert np.isfinite(exp[k][exp["tilesTouched"] != 0]).all(), f"oracle {k}: not a finite-input case any more"
    r, img = _run(scene, cam, bg)
    _compare_all(r, img, exp, n)


def test_screen_filling_needles_at_4k_take_the_unfiltered_path():
    """The blend's strip filter evaluates the power a second, fused way and allows for 0.25 of disagreement with the reference's
    evaluation: enough for terms up to 4e6. A needle across a 3840 x 2160 frame has conic entries near 3.3 and |d| in the
    thousands — terms of 5e7, rounding of +-3 in the power: such records are staged with a zero filter conic (blend_core.hpp,
    stage_and_composite) and the reference-order arithmetic decides alone. Every output as the oracle's."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    n, w, h = 400, 3840, 2160
    scene = scenes.garden_like_scene(n, seed=9100)
    scene["means3D"][:, :3] *= 0.2
    rng = np.random.default_rng(9101)
    needles = np.arange(0, n, 4)
    scene["scales"][needles, 0] = 8.0                       # tens of thousands of pixels long ...
    scene["scales"][needles, 1:3] = 1e-5                    # ... and far below a pixel wide: only the 0.3 low-pass gives them a width
    ang = rng.uniform(0.0, np.pi, needles.size)             # in the image plane, any direction
    scene["rotations"][needles] = np.stack([np.cos(ang / 2), np.zeros_like(ang), np.zeros_like(ang), np.sin(ang / 2)], 1).astype(np.float32)
    scene["opacities"][needles] = rng.uniform(0.05, 1.0, needles.size).astype(np.float32)
    cam = camera.default_camera(w, h, near=0.01, far=100.0)
    exp = cpu_oracle.forward(scene, cam, (0.1, 0.2, 0.3), threads=8)
    co = exp["conicOpacity"][needles]
    assert (exp["tilesTouched"][needles] > 2000).sum() > 20 and float(np.abs(co[:, :3]).max()) > 1.0      # they do cross the frame
    r, img = _run(scene, cam, (0.1, 0.2, 0.3))
    _compare_all(r, img, exp, n)


@pytest.mark.parametrize("near,extra", [(0.5, 12), (0.5, 1), (1.3, 0), (3.0, 0), (0.01, 0)])
def test_depth_keys_outside_the_main_top_byte(near, extra):
    """The depth order sorts the keys whose top byte is 0x3F (NDC z in [0.5, 1): nearly all of them) in three passes and
    ranks the few others on a side list (radix_sort.hip, depth_side_kernel); with more than 1 024 others it falls back to
    the fourth pass. near 0.5 + a dozen (or one) splats between two and four times the near distance (z < 0.5): the side way, low keys in front;
    near 1.3: half of the scene below z = 0.5, the fallback; near 3.0: nothing BUT others (no side way either); near 0.01: no others at all. Lists, ranges, pixels as the oracle's."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    n, w, h = 3000, 320, 200
    scene = scenes.garden_like_scene(n, seed=8100)
    scene["means3D"][:, :3] *= 0.25
    scene["scales"][:, :3] *= 2.0
    if extra:
        rng = np.random.default_rng(8101)
        sel = rng.choice(n, extra, replace=False)
        d = rng.uniform(near * 2.2, near * 3.6, extra)                   # NDC z (glm: [-1, 1]) is 0 at ~2 near and 0.5 at ~4 near
        scene["means3D"][sel, 0] = rng.uniform(-0.05, 0.05, extra) * d
        scene["means3D"][sel, 1] = rng.uniform(-0.05, 0.05, extra) * d
        scene["means3D"][sel, 2] = -5.0 + d
        scene["scales"][sel, :3] = 0.01
    cam = camera.default_camera(w, h, near=near, far=100.0)
    exp = cpu_oracle.forward(scene, cam, (0.0, 0.0, 0.0))
    vis = exp["tilesTouched"] != 0
    top = exp["depths"].view(np.uint32)[vis] >> 24
    others = int((top != 0x3F).sum())
    nvis = int(vis.sum())
    assert {(0.5, 12): 2 <= others <= 16, (0.5, 1): 1 <= others <= 3, (1.3, 0): 1024 < others < nvis, (3.0, 0): others == nvis,
            (0.01, 0): others == 0}[(near, extra)], (others, nvis)
    r, img = _run(scene, cam)
    _compare_all(r, img, exp, n)


@pytest.mark.parametrize("fused,records", [("1", "1"), ("1", "0"), ("0", "1")])
@pytest.mark.parametrize("case", ["top_byte_side_way", "top_byte_fallback", "anisotropic", "ragged_tail", "trained_like_far"])
def test_first_depth_pass_without_the_compaction(case, fused, records, library_env):
    """Scenes beyond 16 M Gaussians skip the compaction of the visible keys: the first depth pass reads the per-Gaussian keys
    itself and leaves out what has no tile (onesweep_kernel, DROP; GSR_FUSED_DEPTH=1 forces that route at any size), and the
    (key, index, rectangle) triples travel between the passes as 12-byte records (REC; GSR_DEPTH_RECORDS=0: as three arrays;
    1: also behind the compaction, where the default keeps the arrays).
    Same lists, ranges and pixels as the oracle's — with keys on the side list, with the four-pass fallback, with a last
    tile that is mostly padding, and with culled Gaussians between the visible ones."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    library_env(GSR_FUSED_DEPTH=fused, GSR_DEPTH_RECORDS=records)
    bg = (0.1, 0.2, 0.3)
    if case in ("top_byte_side_way", "top_byte_fallback"):
        n, w, h = 3000, 320, 200
        near = 0.5 if case == "top_byte_side_way" else 1.3
        scene = scenes.garden_like_scene(n, seed=8100)
        scene["means3D"][:, :3] *= 0.25
        scene["scales"][:, :3] *= 2.0
        if case == "top_byte_side_way":
            rng = np.random.default_rng(8101)
            sel = rng.choice(n, 12, replace=False)
            d = rng.uniform(near * 2.2, near * 3.6, 12)
            scene["means3D"][sel, 0] = rng.uniform(-0.05, 0.05, 12) * d
            scene["means3D"][sel, 1] = rng.uniform(-0.05, 0.05, 12) * d
            scene["means3D"][sel, 2] = -5.0 + d
            scene["scales"][sel, :3] = 0.01
        cam = camera.default_camera(w, h, near=near, far=100.0)
    elif case == "anisotropic":
        n = 20000
        scene = scenes.garden_like_scene(n, seed=11)
        scene["means3D"][:, :3] *= 0.25
        cam = camera.default_camera(333, 257, near=0.05, far=50.0)
    elif case == "ragged_tail":
        n = 8192 + 37                                   # (a second tile of 37 keys; half the scene behind the camera: culled)
        scene = scenes.garden_like_scene(n, seed=12)
        scene["means3D"][:, :3] *= 0.5
        cam = camera.default_camera(200, 120, near=0.05, far=50.0, position=(0.0, 0.0, -0.5))
    else:
        n = 60_000
        scene = scenes.trained_like(n, seed=46)
        cam = camera.default_camera(640, 368, near=0.02, far=120.0, position=(0.0, 0.0, -48.0))
    exp = cpu_oracle.forward(scene, cam, bg, threads=8)
    vis = exp["tilesTouched"] != 0
    assert 0 < int(vis.sum()) < n                        # (some Gaussians to leave out)
    if case == "top_byte_side_way":
        others = int(((exp["depths"].view(np.uint32)[vis] >> 24) != 0x3F).sum())
        assert 2 <= others <= 16
    r, img = _run(scene, cam, bg)
    _compare_all(r, img, exp, n)


def test_four_waves_per_tile_blend_matches_one_wave_per_tile():
    """Calls with few tiles blend with four waves per tile (one 16 x 4 strip each) unless the staged records are
    counted; both forms must give the same pixels, nContrib and finalT bit for bit."""
    from gsrast_amd import camera, scenes
    scene = scenes.garden_like_scene(5000, seed=77)
    scene["means3D"][:, :3] *= 0.25
    cam = camera.default_camera(333, 257, near=0.05, far=50.0)
    torch, SplatRasterizer = _gpu()
    r = SplatRasterizer(333, 257, background=(0.3, 0.2, 0.1))
    r.configure_from_scene(scene)
    one = r.draw(cam, plan=PLAN, count_staged=True).clone()
    st1 = {k: v.clone() for k, v in r.map_image_state().items()}
    four = r.draw(cam, plan=PLAN)
    st4 = r.map_image_state()
    assert torch.equal(one, four)
    assert torch.equal(st1["nContrib"], st4["nContrib"]) and torch.equal(st1["finalT"], st4["finalT"])


def test_blend_exponential_is_the_hosts_expf_bit_for_bit():
    """alpha = min(0.99, opacity * exp(power)) (GSCuda.cu:645) feeds the two hard tests of the tile loop and, through T, every
    later record of the pixel. The HIP blend computes that exponential as glibc's expf does (blend_core.hpp, exp_ref), so that
    alpha, T and the decisions are the CPU oracle's bit for bit: 2e7 arguments over the range the blend can meet (the power
    is <= 0 where it is used; lists of faint splats go down to -100), denormal results and -inf included. (A NaN power is
    no candidate of the blend's filter: not an argument the function meets.)"""
    import torch
    from gsrast_amd import _capi
    from oracle import cpu_oracle
    L = _capi.lib()
    rng = np.random.default_rng(77)
    x = np.concatenate([
        -rng.uniform(0.0, 6.0, 12_000_000), -rng.uniform(0.0, 104.0, 6_000_000), rng.uniform(0.0, 0.5, 1_000_000),
        -np.exp(rng.uniform(-40.0, 2.0, 1_000_000)),
        np.array([0.0, -0.0, -5.5412636, -5.54126, -87.3, -88.0, -100.0, -103.9, -104.0, -104.5, -1e30, -np.inf, 1e-30, -1e-30]),
    ]).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    out = torch.empty_like(xd)
    _capi.check(L.gsr_blend_expf(x.size, xd.data_ptr(), out.data_ptr(), None), "gsr_blend_expf")
    torch.cuda.synchronize()
    got, exp = out.cpu().numpy(), cpu_oracle.expf(x)
    same = got.view(np.uint32) == exp.view(np.uint32)
    assert bool(same.all()), (int((~same).sum()), x[~same][:5], got[~same][:5], exp[~same][:5])


def test_footprint_test_never_drops_a_record_that_lights_a_pixel():
    """The blend drops a record at staging when it has proven that none of the tile's pixels can pass alpha >= 1/255
    (blend_core.hpp). Brute force over random (record, tile) pairs — thin, rotated, huge, faint, opaque, far-away and
    degenerate conics; centres inside, near and far from the tile; partial tiles at the image border: wherever the
    library says "misses", the reference's per-pixel arithmetic (float32, GSCuda.cu:626-646) must agree for all 256
    pixels; and the test must not be vacuous (it drops most records that indeed miss)."""
    import torch
    from gsrast_amd import _capi
    L = _capi.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    n = 400_000
    W, H = 1000, 700                                        # 63 x 44 tiles, partial ones on the right / bottom
    tx = torch.randint(0, 63, (n,), generator=g); ty = torch.randint(0, 44, (n,), generator=g)
    # covariance = R diag(s1^2, s2^2) R^T + 0.3 I, conic = its inverse (what preprocess produces), sizes over 4 decades
    s1 = torch.exp(torch.rand(n, generator=g) * 9.0 - 2.0); ratio = torch.exp(-torch.rand(n, generator=g) * 6.0)
    s2 = s1 * ratio
    th = torch.rand(n, generator=g) * 3.14159265
    c, s = torch.cos(th), torch.sin(th)
    a = c * c * s1 * s1 + s * s * s2 * s2 + 0.3; b = c * s * (s1 * s1 - s2 * s2); d = s * s * s1 * s1 + c * c * s2 * s2 + 0.3
    det = a * d - b * b
    conic = torch.stack([d / det, -b / det, a / det], 1).to(torch.float32)
    op = torch.sigmoid(torch.randn(n, generator=g) * 3.0)
    op[:2000] = torch.tensor([0.0, 1.0, 1e-3, 0.999]).repeat(500)            # corner opacities
    op[2000:2100] = 5.0                                                    # opacity above 1 (not a sigmoid's output, but legal input)
    # centres: within a few footprint radii of the tile, on either side of "just reaches it"
    reach = 3.5 * torch.sqrt(torch.maximum(a, d))
    cx = (tx * 16 + 8).float() + (torch.rand(n, generator=g) * 2 - 1) * (0.6 * reach + 16.0)
    cy = (ty * 16 + 8).float() + (torch.rand(n, generator=g) * 2 - 1) * (0.6 * reach + 16.0)
    xy = torch.stack([cx, cy], 1).to(torch.float32)
    co = torch.cat([conic, op[:, None].float()], 1).contiguous()
    # degenerate records must be KEPT whatever they are
    co[2100:2110, 0] = -1.0; co[2110:2120, 1] = 1e9; co[2120:2130, 3] = float("nan"); xy[2130:2140, 0] = float("inf")
    tile = torch.stack([tx, ty], 1).to(torch.int32)
    xy_d, co_d, tile_d = xy.to(dev), co.to(dev), tile.to(dev)
    misses = torch.zeros(n, dtype=torch.uint8, device=dev)
    rc = L.gsr_footprint_misses_tile(n, xy_d.data_ptr(), co_d.data_ptr(), tile_d.data_ptr(), W, H, misses.data_ptr(), None)
    _capi.check(rc, "gsr_footprint_misses_tile")
    torch.cuda.synchronize()
    # brute force, float32, the reference's operation order; pixels outside the image do not exist
    px = (tile_d[:, 0:1] * 16 + torch.arange(16, device=dev)[None, :]).float()          # [n, 16]
    py = (tile_d[:, 1:2] * 16 + torch.arange(16, device=dev)[None, :]).float()
    lit = torch.zeros(n, dtype=torch.bool, device=dev)
    A, B, Cc, O = co_d[:, 0:1], co_d[:, 1:2], co_d[:, 2:3], co_d[:, 3:4]
    for r in range(16):
        dx = xy_d[:, 0:1] - px                                                            # [n, 16]
        dy = (xy_d[:, 1:2] - py[:, r:r + 1]).expand(-1, 16)
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        alpha = torch.minimum(torch.full_like(power, 0.99), O * torch.exp(power))
        alpha = torch.where(torch.isnan(O * torch.exp(power)), torch.full_like(power, 0.99), alpha)   # fminf(0.99, NaN) = 0.99
        ok = ~(power > 0) & ~(alpha < 1.0 / 255.0) & (px < W) & (py[:, r:r + 1] < H)
        lit |= ok.any(1)
    m = misses.bool()
    wrong = m & lit
    assert int(wrong.sum()) == 0, (int(wrong.sum()), xy_d[wrong][:3], co_d[wrong][:3], tile_d[wrong][:3])
    assert not bool(m[2100:2140].any())                                    # degenerate records are kept
    dark = ~lit
    assert int(dark.sum()) > 50_000 and int(lit.sum()) > 50_000, (int(dark.sum()), int(lit.sum()))
    assert int((m & dark).sum()) >= 0.9 * int(dark.sum()), (int((m & dark).sum()), int(dark.sum()))   # and it is sharp
