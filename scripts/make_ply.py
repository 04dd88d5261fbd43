"""Writes a scene of the generator (gsrast_amd/scenes.py) as a .ply in the reference's format (apps/gsrast/SplatData.hpp:17-25,
SplatData.cpp:114-156: 62 little-endian floats per vertex — position 3, normal 3, f_dc 3 + f_rest 45, opacity 1, scale 3,
rotation 4 —, property names never parsed) with PRE-ACTIVATION values, as a trained file holds them: log scales, logit
opacities, raw (unnormalised) quaternions. The loader (gsr_ply_activate, SplatData.cpp:50-54) takes them back.
  python scripts/make_ply.py OUT.ply [trained_like|garden_like] [splats, default 5834784]
The garden .ply itself is not available offline; scenes.trained_like at its splat count (1.45 GB) stands in for it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def scene_to_ply(path, scene, quat_gain_seed=11):
    """scene: activated SoA (scenes.py). Returns the bytes written."""
    from gsrast_amd import ply
    n = int(scene["means3D"].shape[0])
    opac = np.clip(scene["opacities"].astype(np.float64), 1e-7, 1.0 - 1e-7)
    # a trained file's quaternions are not unit: scaled by a per-splat factor here, which the loader's normalisation removes
    gain = np.random.default_rng(quat_gain_seed).uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    ply.write_ply(path, scene["means3D"][:, :3], scene["shs"], np.log(opac / (1.0 - opac)).astype(np.float32),
                  np.log(scene["scales"][:, :3].astype(np.float64)).astype(np.float32), scene["rotations"] * gain)
    return os.path.getsize(path)


if __name__ == "__main__":
    from gsrast_amd import scenes
    out = sys.argv[1]
    which = sys.argv[2] if len(sys.argv) > 2 else "trained_like"
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 5_834_784
    sc = scenes.trained_like(n, seed=45) if which == "trained_like" else scenes.garden_like_scene(n, seed=43)
    size = scene_to_ply(out, sc)
    print(f"wrote {out}: {n} splats ({which}), {size / 1e9:.3f} GB")
