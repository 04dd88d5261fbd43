"""CPU: closed-form known-answer tests of the oracle (SURVEY.md §8c iv). These pin the CPU
restatement to the reference's formulas (apps/gsrast/gscuda/GSCuda.cu) independently of any
second implementation."""
import math

import numpy as np

from helpers import single_gaussian_scene
from gsrast_amd import camera
from oracle import cpu_oracle


def test_single_isotropic_gaussian_on_the_optical_axis():
    W = H = 128
    s, d, o, dc = 0.1, 5.0, 0.8, (1.0, 0.5, -0.5)
    cam = camera.default_camera(W, H)
    scene = single_gaussian_scene(pos=(0, 0, 0), scale=s, opacity=o, dc=dc, n=2)   # two, so R > 1 draws
    scene["means3D"][1, :3] = (0.0, 0.0, -60.0)                                       # second one culled
    bg = (0.1, 0.2, 0.3)
    out = cpu_oracle.forward(scene, cam, bg)
    f = H / (2.0 * cam.tan_fovy)
    var = (f * s / d) ** 2 + 0.3                       # GSCuda.cu:213-227 with one focal for both axes
    assert np.allclose(out["means2D"][0], (W / 2, H / 2), atol=1e-3)          # no half-pixel shift (:342)
    assert np.allclose(out["conicOpacity"][0], (1 / var, 0.0, 1 / var, o), rtol=1e-4, atol=1e-6)
    assert out["radii"][0] == math.ceil(3.0 * math.sqrt(var))                 # :338-341
    assert tuple(out["rects"][0]) == (math.ceil(3 * math.sqrt(var)), math.ceil(3 * var))   # sic :352
    assert out["radii"][1] == 0 and out["tilesTouched"][1] == 0               # behind the camera
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    power = -0.5 * ((xs - W / 2) ** 2 + (ys - H / 2) ** 2) / var
    alpha = np.minimum(0.99, o * np.exp(power))
    alpha[alpha < 1 / 255] = 0.0
    for c in range(3):
        want = (0.5 + 0.4 * dc[c]) * alpha + (1 - alpha) * bg[c]                # :362-366, :661, :673
        assert np.abs(out["out_color"][c] - want).max() < 2e-5
    assert np.abs(out["finalT"] - (1 - alpha)).max() < 2e-6
    assert set(np.unique(out["nContrib"])) <= {0, 1}


def test_two_coincident_centres_blend_front_to_back():
    W = H = 64
    cam = camera.default_camera(W, H)
    scene = single_gaussian_scene(pos=(0, 0, 0), scale=0.2, opacity=0.6, n=2)
    scene["means3D"][0, 2] = 1.0          # farther (camera sits at z = -5 looking along +z)
    scene["means3D"][1, 2] = -1.0         # nearer
    scene["shs"][0, :3] = (1.0, 1.0, 1.0)     # far: colour 0.9
    scene["shs"][1, :3] = (-1.0, -1.0, -1.0)  # near: colour 0.1
    out = cpu_oracle.forward(scene, cam)
    cy, cx = H // 2, W // 2
    k = out["values"][out["ranges"][(cy // 16) * 4 + cx // 16][0]:out["ranges"][(cy // 16) * 4 + cx // 16][1]]
    assert list(k) == [1, 0]                                                  # nearer first (smaller NDC z)
    a_near = min(0.99, 0.6 * math.exp(0.0))
    a_far = a_near
    want = 0.1 * a_near + 0.9 * a_far * (1 - a_near)
    assert abs(out["out_color"][0, cy, cx] - want) < 1e-5
    assert abs(out["finalT"][cy, cx] - (1 - a_near) * (1 - a_far)) < 1e-6
    assert out["nContrib"][cy, cx] == 2


def test_opaque_stack_stops_at_transmittance_cutoff():
    W = H = 32
    cam = camera.default_camera(W, H)
    n = 40
    scene = single_gaussian_scene(pos=(0, 0, 0), scale=1.0, opacity=0.9, n=n)
    scene["means3D"][:, 2] = np.linspace(-1.0, 1.0, n)
    out = cpu_oracle.forward(scene, cam)
    # centre pixel: alpha = 0.9 each; T after k layers = 0.1^k; layer 4 would give 1e-4 < 0.001 -> stop (:653)
    assert abs(out["finalT"][16, 16] - 1e-3) < 1e-6
    assert out["nContrib"][16, 16] == 3


def test_ties_keep_ascending_index_and_culls():
    W = H = 64
    cam = camera.default_camera(W, H)
    scene = single_gaussian_scene(pos=(0.1, 0.1, 0.0), scale=0.1, opacity=0.5, n=5)
    scene["means3D"][3, :3] = (50.0, 0.0, 0.0)        # |ndc.x| > 1.3 -> culled (:306)
    scene["means3D"][4, :3] = (0.0, 0.0, -4.995)      # NDC z < 0 -> culled
    out = cpu_oracle.forward(scene, cam)
    assert list(out["radii"][3:] > 0) == [False, False]
    keys, vals = out["keys"], out["values"]
    same = keys[1:] == keys[:-1]
    assert same.any() and (vals[1:][same] > vals[:-1][same]).all()


def test_zero_and_one_instance_edge_cases():
    cam = camera.default_camera(64, 64)
    none = cpu_oracle.forward(single_gaussian_scene(pos=(0, 0, -50.0), n=2), cam, out_init=np.full((3, 64, 64), 0.25))
    assert none["num_rendered"] == 0 and (none["out_color"] == 0.25).all()       # :775-778
    one = cpu_oracle.forward(single_gaussian_scene(pos=(0.5178, -0.5178, 0.0), scale=0.001, n=1), cam, (0.2, 0.3, 0.4))
    assert one["num_rendered"] == 1
    assert one["ranges"].max() == 0                                               # R == 1 quirk (:515-537)
    assert np.allclose(one["out_color"][1], 0.3)


def test_partial_tiles_only_write_inside_pixels():
    W, H = 40, 24          # 3 x 2 tiles, the last column / row half empty
    cam = camera.default_camera(W, H)
    scene = single_gaussian_scene(pos=(0, 0, 0), scale=0.5, opacity=0.7, n=3)
    out = cpu_oracle.forward(scene, cam, (0.5, 0.5, 0.5))
    assert out["out_color"].shape == (3, H, W) and np.isfinite(out["out_color"]).all()
    assert (out["finalT"] < 1.0).any()
