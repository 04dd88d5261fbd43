"""The .ply loader row: header quirks on the CPU, activations on the GPU."""
import os

import numpy as np
import pytest

from gsrast_amd import ply
from oracle import ply_oracle


def _random_scene(n, seed=3):
    rng = np.random.default_rng(seed)
    return dict(position=rng.normal(0, 1.5, (n, 3)).astype(np.float32), sh=rng.normal(0, 0.4, (n, 48)).astype(np.float32),
                opacity_logit=rng.normal(0, 3, n).astype(np.float32), log_scale=rng.normal(-3.0, 0.8, (n, 3)).astype(np.float32),
                rotation=rng.normal(0, 1, (n, 4)).astype(np.float32), normal=rng.normal(0, 1, (n, 3)).astype(np.float32))


def test_header_is_read_the_way_the_reference_reads_it(tmp_path):
    p = str(tmp_path / "a.ply")
    ply.write_ply(p, **_random_scene(37))
    n, off = ply.parse_header(p)
    assert n == 37 and os.path.getsize(p) - off == 37 * 248
    # the count is the third token of the third line, whatever the first two tokens say
    q = str(tmp_path / "b.ply")
    with open(q, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nfoo bar 5 trailing\ncomment anything\nend_header\n" + b"\0" * (5 * 248))
    assert ply.parse_header(q) == (5, os.path.getsize(q) - 5 * 248)
    with open(q, "wb") as f:
        f.write(b"ply\nformat x\nelement vertex 5\nno terminator\n")
    with pytest.raises(ValueError):
        ply.parse_header(q)
    o = ply_oracle.load(p)
    # f_rest is channel-major in the file; "coefficient_major" regroups it as [15][3] behind the DC triple
    oc = ply_oracle.load(p, sh_layout="coefficient_major")
    assert np.array_equal(oc["shs"][:, :3], o["shs"][:, :3])
    for k in range(1, 16):
        for ch in range(3):
            assert np.array_equal(oc["shs"][:, 3 * k + ch], o["shs"][:, 3 + 15 * ch + (k - 1)])
    assert o["means3D"].shape == (37, 4) and (o["means3D"][:, 3] == 1).all()
    assert np.allclose(np.linalg.norm(o["rotations"], axis=1), 1.0, atol=1e-6)


@pytest.mark.gpu
def test_gpu_activations_match_the_cpu_loader(tmp_path):
    p = str(tmp_path / "scene.ply")
    n = 100_003                                            # not a multiple of the wave / block size
    ply.write_ply(p, **_random_scene(n, seed=9))
    want = ply_oracle.load(p)
    got = ply.load_ply(p)
    assert np.array_equal(got["means3D"].cpu().numpy(), want["means3D"])
    assert np.array_equal(got["shs"].cpu().numpy(), want["shs"])
    assert np.array_equal(got["rotations"].cpu().numpy(), want["rotations"])          # IEEE sqrt / divide
    for k in ("scales", "opacities"):                      # expf: device vs numpy, a few ulp at most
        a, b = got[k].cpu().numpy(), want[k]
        assert np.abs(a - b).max() <= 4e-7 * np.abs(b).max() + 1e-7 * 0 + 3 * np.spacing(np.abs(b)).max(), k
    assert np.allclose(got["bbox_min"].cpu().numpy(), want["bbox_min"]) and np.allclose(got["center"].cpu().numpy(), want["center"], atol=1e-5)
    with open(p, "r+b") as f:                               # a truncated file is rejected (SplatData.cpp:147-152)
        f.truncate(os.path.getsize(p) - 100)
    with pytest.raises(ValueError):
        ply.load_ply(p)


@pytest.mark.gpu
def test_scene_loaded_from_ply_renders_like_the_in_memory_scene(tmp_path):
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    sc = scenes.isotropic_scene(1000, 42)
    p = str(tmp_path / "iso.ply")
    opac = np.clip(sc["opacities"].astype(np.float64), 1e-6, 1 - 1e-6)
    ply.write_ply(p, sc["means3D"][:, :3], sc["shs"], np.log(opac / (1 - opac)), np.log(sc["scales"][:, :3]), sc["rotations"])
    loaded = ply.load_ply(p)
    cam = camera.default_camera(128, 128)
    a, b = SplatRasterizer(128, 128), SplatRasterizer(128, 128)
    a.configure_from_scene(sc)
    b.configure_from_scene({k: loaded[k] for k in ("means3D", "scales", "rotations", "opacities", "shs")})
    ia, ib = a.draw(cam).cpu().numpy(), b.draw(cam).cpu().numpy()
    assert np.abs(ia - ib).max() <= 2e-3        # log/exp round trip of scales and opacities moves a few thresholds
    assert abs(a.last_num_rendered - b.last_num_rendered) <= a.last_num_rendered // 200


@pytest.mark.gpu
def test_ply_with_higher_order_sh_renders_under_the_inria_profile(tmp_path):
    """A .ply with random f_rest, loaded with sh_layout="coefficient_major", drawn with the upstream semantics
    (degree 3) against oracle/inria_np.py fed by the CPU loader: the f_rest transposition must be the right one
    (loaded in file order the colours come out visibly wrong)."""
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    from oracle import inria_np
    sc = scenes.garden_like_scene(3000, seed=5)
    sc["means3D"][:, :3] *= 0.25
    rng = np.random.default_rng(5)
    file_sh = rng.normal(0, 0.3, (3000, 48)).astype(np.float32)          # as stored: f_dc, then f_rest channel-major
    p = str(tmp_path / "sh.ply")
    opac = np.clip(sc["opacities"].astype(np.float64), 1e-6, 1 - 1e-6)
    ply.write_ply(p, sc["means3D"][:, :3], file_sh, np.log(opac / (1 - opac)), np.log(sc["scales"][:, :3]), sc["rotations"])
    want_scene = ply_oracle.load(p, sh_layout="coefficient_major")
    got = ply.load_ply(p, sh_layout="coefficient_major")
    assert np.array_equal(got["shs"].cpu().numpy(), want_scene["shs"])
    assert np.array_equal(ply.load_ply(p)["shs"].cpu().numpy(), file_sh)
    cam = camera.default_camera(200, 120, near=0.05, far=50.0)
    bg = (0.1, 0.2, 0.3)
    # the oracle renders the scene the GPU loader produced (its expf differs from numpy's by an ulp in scales / opacities)
    host = {k: got[k].cpu().numpy() for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    exp = inria_np.forward(host, cam, bg, deg=3)
    r = SplatRasterizer(200, 120, background=bg)
    r.configure_from_scene({k: got[k] for k in host})
    img = r.draw(cam, semantics="inria", sh_degree=3).cpu().numpy()
    assert r.last_num_rendered == exp["num_rendered"]
    assert np.abs(img - exp["out_color"]).max() <= 1e-4
    wrong = ply.load_ply(p)                                               # file order fed to the [16][3] reader
    r.configure_from_scene({k: wrong[k] for k in host})
    assert np.abs(r.draw(cam, semantics="inria", sh_degree=3).cpu().numpy() - exp["out_color"]).max() > 1e-2


@pytest.mark.gpu
def test_bench_takes_a_scene_file_when_one_is_there(tmp_path):
    """bench.py --ply PATH (or $GSR_GARDEN_PLY): the scene comes through the reference's loader semantics
    (SplatData.cpp:114-156) and the app's camera rule (GSRastWindow.cpp:30-36), the line says "data": "ply:<file>" and
    carries the CPU baseline of that same frame; a path that does not exist leaves the synthetic stand-in in place."""
    import json
    import subprocess
    import sys
    from helpers import ROOT
    p = str(tmp_path / "garden_small.ply")
    sc = _random_scene(40_000, seed=21)
    sc["position"] *= 0.4
    ply.write_ply(p, **sc)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--width", "640", "--height", "368",
           "--ply", p]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["data"] == "ply:garden_small.ply" and d["config"]["splats"] == 40_000 and d["value"] > 0
    assert "ply:garden_small.ply" in d["config"]["workload"] and d["config"]["num_rendered"] > 0
    assert d["cpu_baseline"]["comparable"] is True and d["cpu_baseline"]["value"] > 0
    assert "pose_outside" not in d                      # the extras belong to the default workload
    env = dict(os.environ, GSR_GARDEN_PLY=str(tmp_path / "absent.ply"))
    out = subprocess.run(cmd[:-2] + ["--splats", "20000", "--no-cpu-baseline", "--no-extras"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["data"] == "synthetic" and "does not exist" in out.stderr


@pytest.mark.gpu
def test_full_size_scene_file_through_the_loader(tmp_path):
    """BASELINE config 2's real code path at its size: 5 834 784 splats written as a reference-format file (1.45 GB, pre-activation
    values: scripts/make_ply.py) -> gsr_ply_parse_header -> gsr_ply_activate -> forward. The loaded SoA against the CPU loader
    (oracle/ply_oracle.py, SplatData.cpp:28-66) value by value: positions, SH and the normalised quaternions bit for bit,
    exp / sigmoid within the device's expf (stated: scales within 2 ulp of the CPU loader's, opacities — 1 / (1 + e) of it —
    within 4), and the frame of the loaded scene against the frame of the
    very same SoA configured from host memory: bit for bit."""
    import sys
    import torch
    from helpers import ROOT
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import make_ply
    from gsrast_amd import camera, scenes
    from gsrast_amd.rasterizer import SplatRasterizer
    n = 5_834_784
    p = str(tmp_path / "trained_like.ply")
    scene = scenes.trained_like(n, seed=45)
    assert make_ply.scene_to_ply(p, scene) > n * 248
    assert ply.parse_header(p)[0] == n
    got = ply.load_ply(p)
    want = ply_oracle.load(p)
    for k in ("means3D", "shs", "rotations"):
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    for k in ("scales", "opacities"):
        a, b = got[k].cpu().numpy(), want[k]
        ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
        assert int(ulp.max()) <= (2 if k == "scales" else 4), (k, int(ulp.max()))
    del want
    w, h = 1920, 1080
    span = float((got["bbox_max"] - got["bbox_min"]).max().item())
    cam = camera.default_camera(w, h, near=0.001 * span, far=span, position=(0.0, 0.0, -5.0))        # GSRastWindow.cpp:30-36
    keys = ("means3D", "scales", "rotations", "opacities", "shs")
    a = SplatRasterizer(w, h)
    a.configure_from_scene({k: got[k] for k in keys})
    ia = a.draw(cam).clone()
    b = SplatRasterizer(w, h)
    b.configure_from_scene({k: got[k].cpu().numpy() for k in keys})
    ib = b.draw(cam)
    assert a.last_num_rendered == b.last_num_rendered > 10_000_000
    assert torch.equal(ia.view(torch.int32), ib.view(torch.int32))
    # and the scene the file was written from is what came back: positions exactly, the activations through float32 log / exp
    assert np.array_equal(got["means3D"].cpu().numpy(), scene["means3D"])
    rel = np.abs(got["scales"].cpu().numpy()[:, :3] / scene["scales"][:, :3] - 1.0).max()
    assert rel <= 2e-6, rel
