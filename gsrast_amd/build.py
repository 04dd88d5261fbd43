"""In-tree build of the HIP library: gsrast_amd/csrc/*.hip -> gsrast_amd/lib/libgsrast_amd.so.

gfx950 only, one hipcc invocation per translation unit (cached by mtime), then one link.
The .so stays in the tree (git-ignored) so it travels with the source snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libgsrast_amd.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: the preprocess / blend arithmetic keeps the reference's float32 operation
# order (no fused multiply-add), which is what makes integer outputs reproducible bit for bit.
COMMON = (["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
          + os.environ.get("GSR_DEFINES", "").split())      # tuning experiments: GSR_DEFINES="-DX=1 ..."


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime() -> float:
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    paths.append(os.path.join(HERE, "..", "include", "gsrast_amd.h"))
    return max(os.path.getmtime(p) for p in paths)


def _compile(src: str, force: bool, obj_dir: str = OBJ_DIR) -> str:
    obj = os.path.join(obj_dir, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    stale = (force or not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(spath)
             or os.path.getmtime(obj) < _headers_mtime())
    if stale:
        cmd = [HIPCC] + COMMON + ["-c", spath, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False, tag: str = "") -> str:
    """tag: experiments only — builds lib/libgsrast_amd_<tag>.so from its own object directory
    (always from scratch, with whatever GSR_DEFINES is set), selected at run time by GSR_LIB_TAG."""
    obj_dir = OBJ_DIR + ("_" + tag if tag else "")
    lib_path = LIB_PATH if not tag else os.path.join(LIB_DIR, f"libgsrast_amd_{tag}.so")
    force = force or bool(tag)
    os.makedirs(obj_dir, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, obj_dir), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(lib_path) or os.path.getmtime(lib_path) < newest:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_path] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", lib_path)
    return lib_path


if __name__ == "__main__":
    _tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else ""
    build(force="--force" in sys.argv, verbose=True, tag=_tag)
