"""Point-splat path (gscuda::forwardPoints, reference GSCuda.cu:26-155): numpy restatement on CPU, the HIP
path against it on the GPU (bit for bit: integer pixel choice, float32 colours and depths)."""
import ctypes as C

import numpy as np
import pytest

from oracle import points_np


def _scene(n, seed):
    from gsrast_amd import scenes
    sc = scenes.garden_like_scene(n, seed=seed)
    sc["means3D"][:, :3] *= 0.3
    return sc


def test_points_oracle_basics():
    from gsrast_amd import camera
    cam = camera.default_camera(64, 48)
    sc = _scene(400, 3)
    sc["means3D"][0, :3] = (0.0, 0.0, -4.5)          # on the optical axis, nearest of all -> centre pixel
    sc["means3D"][1, :3] = (0.0, 0.0, -4.0)          # behind it on the same ray: loses the depth test
    sc["means3D"][2, :3] = (0.0, 0.0, -50.0)         # behind the camera: culled
    out, depth = points_np.forward_points(sc["means3D"][:, :3], sc["shs"], cam.proj, 64, 48, (0.1, 0.2, 0.3))
    assert out.shape == (3, 48, 64) and depth.shape == (48, 64)
    assert np.allclose(out[:, 24, 32], 0.4 * sc["shs"][0, :3] + 0.5)
    empty = depth == 1.0
    assert empty.any() and (~empty).any()
    assert np.allclose(out[:, empty], np.array([0.1, 0.2, 0.3], np.float32)[:, None])
    assert (depth >= 0).all() and (depth <= 1).all()


def test_points_chunk_layout_matches_the_reference_carve_order():
    """pc::ImageState::fromChunk (AuxBuffer.cu:31-39): depth f32[P], outColor f32[3P], defaultDepth f32[1], each 128-B aligned."""
    from gsrast_amd import _capi
    L = _capi.lib()
    st = _capi.PointsImageState()
    base, P = 1 << 20, 1000
    end = L.gsr_points_image_from_chunk(base, P, C.byref(st))
    assert st.depth == base
    assert st.out_color == base + 4096                       # 4000 -> next multiple of 128
    assert st.default_depth == st.out_color + 12032          # 12000 -> 12032
    assert st.winner == st.default_depth + 128
    assert end == st.winner + 8 * P
    assert L.gsr_required_points_image(P) == end - base


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,n,seed", [(64, 48, 400, 3), (333, 257, 200_000, 5), (1920, 1080, 2_000_000, 9)])
def test_points_hip_matches_oracle(w, h, n, seed):
    import torch
    from gsrast_amd import camera
    from gsrast_amd.rasterizer import SplatRasterizer
    cam = camera.default_camera(w, h)
    sc = _scene(n, seed)
    bg = (0.1, 0.2, 0.3)
    exp_out, exp_depth = points_np.forward_points(sc["means3D"][:, :3], sc["shs"], cam.proj, w, h, bg)
    r = SplatRasterizer(w, h, background=bg)
    r.configure_from_scene(sc)
    img = r.draw_points(cam).cpu().numpy()
    st = r.map_points_image_state()
    assert np.array_equal(st["depth"].cpu().numpy(), exp_depth)
    assert np.array_equal(img, exp_out)
    assert np.array_equal(st["outColor"].cpu().numpy(), exp_out)
    assert len(r.binning.calls) == 0 and r.geom.calls[-1] == 32       # GSCuda.cu:128-134
    again = r.draw_points(cam).cpu().numpy()
    assert np.array_equal(again, img)                                  # deterministic, unlike the reference's race
