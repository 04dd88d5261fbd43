"""CPU: the C++ oracle and the independent numpy restatement against the committed fixture."""
import numpy as np

from helpers import FLT_KEYS, INT_KEYS, load_golden
from oracle import cpu_oracle, oracle_np


def test_cpp_oracle_matches_golden():
    scene, cam, bg, exp = load_golden()
    got = cpu_oracle.forward(scene, cam, bg)
    for k in INT_KEYS:
        assert np.array_equal(got[k], exp[k]), k
    for k in FLT_KEYS:
        assert np.abs(got[k].astype(np.float64) - exp[k]).max() <= 1e-6, k
    assert got["num_rendered"] == exp["num_rendered"] == 8593
    assert got["records_staged"] == exp["records_staged"]


def test_numpy_restatement_matches_golden():
    scene, cam, bg, exp = load_golden()
    got = oracle_np.forward(scene, cam, bg)
    for k in INT_KEYS:
        assert np.array_equal(got[k], exp[k]), k
    for k in FLT_KEYS:
        assert np.abs(got[k].astype(np.float64) - exp[k]).max() <= 1e-6, k


def test_threaded_blend_is_identical():
    scene, cam, bg, _ = load_golden()
    a = cpu_oracle.forward(scene, cam, bg, threads=1)
    b = cpu_oracle.forward(scene, cam, bg, threads=4)
    for k in ("out_color", "finalT", "nContrib"):
        assert np.array_equal(a[k], b[k]), k
    assert a["records_staged"] == b["records_staged"]


def test_oracles_agree_on_anisotropic_non_multiple_of_16():
    """Rotated anisotropic splats, 200x120 image (last tile column/row partly outside)."""
    from gsrast_amd import camera, scenes
    scene = scenes.garden_like_scene(3000, seed=7)
    scene["means3D"][:, :3] *= 0.25
    cam = camera.default_camera(200, 120, near=0.05, far=50.0)
    a = cpu_oracle.forward(scene, cam, (0.1, 0.2, 0.3))
    b = oracle_np.forward(scene, cam, (0.1, 0.2, 0.3))
    for k in INT_KEYS:
        assert np.array_equal(a[k], b[k]), k
    for k in FLT_KEYS:
        assert np.abs(a[k].astype(np.float64) - b[k]).max() <= 2e-6, k
    assert a["num_rendered"] > 0 and a["records_staged"] == b["records_staged"]


def test_radius_rect_path_agrees():
    scene, cam, bg, _ = load_golden()
    a = cpu_oracle.forward(scene, cam, bg, use_rects=False)
    b = oracle_np.forward(scene, cam, bg, use_rects=False)
    for k in ("radii", "tilesTouched", "keys", "values", "ranges", "nContrib"):
        assert np.array_equal(a[k], b[k]), k
    assert np.abs(a["out_color"] - b["out_color"]).max() <= 1e-6


def test_threaded_oracle_gives_the_single_thread_result():
    """bench.py's cpu_baseline runs the oracle on all host cores: preprocess, key emission, the stable sort and the tile
    loop must give, bit for bit, what the single-thread restatement gives."""
    from gsrast_amd import camera, scenes
    from oracle import cpu_oracle
    scene = scenes.garden_like_scene(30_000, seed=9)
    scene["means3D"][:, :3] *= 0.3
    cam = camera.default_camera(320, 200, near=0.05, far=60.0)
    a = cpu_oracle.forward(scene, cam, (0.1, 0.2, 0.3), threads=1)
    b = cpu_oracle.forward(scene, cam, (0.1, 0.2, 0.3), threads=7)
    assert a["num_rendered"] == b["num_rendered"] > 100_000 and a["records_staged"] == b["records_staged"]
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, b[k]), k
