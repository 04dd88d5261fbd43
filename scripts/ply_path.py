"""BASELINE config 2's real code path at full size: a 5 834 784-splat scene FILE in the reference's format -> the reference's
loader semantics (SplatData.cpp:114-156: header, bulk read, exp / normalise / sigmoid — here gsr_ply_parse_header +
gsr_ply_activate on the GPU) -> forward at 1920 x 1080, camera as the app sets it up (GSRastWindow.cpp:30-36). Times every
step and compares the loaded scene with the scene the file was written from.
  python scripts/ply_path.py [PLY, default /tmp/gsr_trained_like.ply — written by scripts/make_ply.py if it is not there]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from gsrast_amd import _capi, camera, ply, scenes
from gsrast_amd.rasterizer import SplatRasterizer
import make_ply

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/gsr_trained_like.ply"
N = 5_834_784
t0 = time.perf_counter()
scene = scenes.trained_like(N, seed=45)
t_gen = time.perf_counter() - t0
if not os.path.exists(path):
    t0 = time.perf_counter()
    size = make_ply.scene_to_ply(path, scene)
    print(f"wrote {path}: {size / 1e9:.3f} GB in {time.perf_counter() - t0:.1f} s (scene generated in {t_gen:.1f} s)")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
# the loader, step by step
t0 = time.perf_counter(); n, off = ply.parse_header(path); t_hdr = time.perf_counter() - t0
t0 = time.perf_counter(); raw = np.fromfile(path, dtype="<f4", offset=off, count=n * ply.RECORD_FLOATS); t_read = time.perf_counter() - t0
t0 = time.perf_counter(); raw_dev = torch.from_numpy(raw).to(dev); torch.cuda.synchronize(); t_up = time.perf_counter() - t0
out = {"means3D": torch.empty((n, 4), device=dev), "scales": torch.empty((n, 4), device=dev), "rotations": torch.empty((n, 4), device=dev),
       "opacities": torch.empty((n,), device=dev), "shs": torch.empty((n, 48), device=dev)}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    rc = _capi.lib().gsr_ply_activate_layout(raw_dev.data_ptr(), n, out["means3D"].data_ptr(), out["scales"].data_ptr(), out["rotations"].data_ptr(),
                                             out["opacities"].data_ptr(), out["shs"].data_ptr(), _capi.GSR_SH_LAYOUT_FILE,
                                             torch.cuda.current_stream(dev).cuda_stream)
    e1.record(); torch.cuda.synchronize()
    _capi.check(rc, "gsr_ply_activate_layout")
t_act = e0.elapsed_time(e1)
print(f"{os.path.basename(path)}: {n} splats, {os.path.getsize(path) / 1e9:.3f} GB")
print(f"load: header {t_hdr * 1e3:.2f} ms, bulk read {t_read * 1e3:.0f} ms ({os.path.getsize(path) / t_read / 1e9:.2f} GB/s from the page cache / disk), "
      f"upload {t_up * 1e3:.0f} ms ({raw.nbytes / t_up / 1e9:.1f} GB/s), gsr_ply_activate {t_act:.3f} ms "
      f"({(raw.nbytes + 4 * n * 61) / (t_act * 1e-3) / 1e9:.0f} GB/s of 248 B read + 244 B written per splat)")
# against the scene the file was written from
def ulps(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
got = {k: v.cpu().numpy() for k, v in out.items()}
print("loaded vs the generator's scene: means3D bit-equal", bool(np.array_equal(got["means3D"], scene["means3D"])),
      "| shs bit-equal", bool(np.array_equal(got["shs"], scene["shs"])),
      f"| scales: max {int(ulps(got['scales'][:, :3], scene['scales'][:, :3]).max())} ulp (log then exp, float32)",
      f"| opacities: max abs diff {float(np.abs(got['opacities'] - scene['opacities']).max()):.2e} (logit then sigmoid)",
      f"| rotations: max abs diff {float(np.abs(got['rotations'] - scene['rotations']).max()):.2e} (scaled, then normalised)")
W, H = 1920, 1080
span = float((out["means3D"][:, :3].max(0).values - out["means3D"][:, :3].min(0).values).max().item())
cam = camera.default_camera(W, H, near=0.001 * span, far=span, position=(0.0, 0.0, -5.0))           # GSRastWindow.cpp:30-36
def frames(sc, title):
    r = SplatRasterizer(W, H, device=dev)
    r.configure_from_scene(sc)
    for _ in range(12):
        r.draw(cam)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); r.draw(cam); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{title}: frame {np.mean(ts):.4f} ms mean / {np.median(ts):.4f} median ({n / np.mean(ts) / 1e3:.0f} Msplats/s), R = {r.last_num_rendered}, plan {r.last_plan}"
          f"{'+lists' if r.last_blend_from_lists else ''}{'+overlap' if r.last_emit_overlapped else ''}{'+deep' if r.last_deep_tiles else ''}")
    return r.out_color.clone(), r.last_num_rendered
img_f, r_f = frames(out, "frame of the LOADED scene, reference's default pose")
img_m, r_m = frames({k: torch.from_numpy(np.ascontiguousarray(scene[k])).to(dev) for k in out}, "frame of the in-memory scene it was written from")
d = (img_f - img_m).abs()
print(f"the two frames: R {r_f} / {r_m}; max abs pixel difference {float(d.max()):.2e}, pixels beyond 1e-4: {int((d.max(0).values > 1e-4).sum())} of {W * H} "
      f"(the float32 log / exp round trip of scales and opacities moves a few thresholds; the loader itself is compared value by value above and in tests/test_ply.py)")
