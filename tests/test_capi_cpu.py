"""CPU: the C-ABI library loads, exports every symbol include/gsrast_amd.h declares, and its
host-only entry points (chunk layout, error strings, getHigherMsb) behave like the reference's
(apps/gsrast/gscuda/AuxBuffer.cu:13-21,44-89; GSCuda.cu:481-502). No compute call is made."""
import ctypes as C
import os
import re

import pytest

from helpers import ROOT

from gsrast_amd import _capi
from oracle import cpu_oracle


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gsrast_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", text)) - {"gsr_alloc_fn"})


def test_library_exports_every_declared_symbol():
    L = _capi.lib()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in gsrast_amd.h but not exported"
    assert set(_capi.SIGNATURES) == set(names), "ctypes table and header disagree"


def test_struct_sizes_match_the_header():
    # gsr_forward refuses a struct whose size differs from the C one; mirror that check host-side
    L = _capi.lib()
    a = _capi.ForwardArgs()
    a.struct_size = C.sizeof(_capi.ForwardArgs) - 4
    assert L.gsr_forward(C.byref(a)) == _capi.GSR_ERR_INVALID_ARG
    a.struct_size = C.sizeof(_capi.ForwardArgs)
    assert L.gsr_forward(C.byref(a)) == _capi.GSR_ERR_INVALID_ARG      # null pointers, still no GPU touched
    assert L.gsr_last_error() == _capi.GSR_ERR_INVALID_ARG
    assert b"invalid" in L.gsr_error_string(_capi.GSR_ERR_INVALID_ARG)


def _obtain(off, size, align=128):
    a = align * ((off + align - 1) // align)
    return a, a + size


@pytest.mark.parametrize("n", [1, 1000, 5_834_784])
def test_geometry_chunk_layout_follows_the_reference_carve_order(n):
    """AuxBuffer.cu:44-63: every array 128-byte aligned, in the reference's order and element sizes."""
    L = _capi.lib()
    st = _capi.GeometryState()
    base = 1 << 20
    end = L.gsr_geometry_from_chunk(base, n, C.byref(st))
    off = base
    p, off = _obtain(off, 4 * n); assert st.tiles_touched == p
    p, off = _obtain(off, st.scan_size); assert st.scanning_space == p
    p, off = _obtain(off, 4 * n); assert st.depths == p
    p, off = _obtain(off, 3 * n); assert st.clamped == p
    p, off = _obtain(off, 4 * n); assert st.internal_radii == p
    p, off = _obtain(off, 8 * n); assert st.means2D == p
    p, off = _obtain(off, 24 * n); assert st.cov3D == p
    p, off = _obtain(off, 16 * n); assert st.conic_opacity == p
    p, off = _obtain(off, 12 * n); assert st.rgb == p
    p, off = _obtain(off, 4 * n); assert st.point_offsets == p
    assert end == off
    assert L.gsr_required_geometry(n) == L.gsr_geometry_from_chunk(0, n, C.byref(st))


def test_image_and_binning_chunk_layout():
    L = _capi.lib()
    im = _capi.ImageState()
    P = 1920 * 1080
    end = L.gsr_image_from_chunk(0, P, C.byref(im))
    # uvec2[P], u32[P], f32[P]; 4P is a multiple of 128 here, so no padding (ctypes maps NULL to None)
    assert ((im.ranges or 0), im.n_contrib, im.accum_alpha) == (0, 8 * P, 12 * P)
    assert end == 16 * P == L.gsr_required_image(P)
    b = _capi.BinningState()
    R = 1_000_003
    end = L.gsr_binning_from_chunk(0, R, C.byref(b))
    off = 0
    p, off = _obtain(off, 8 * R); assert (b.keys_unsorted or 0) == p
    p, off = _obtain(off, 8 * R); assert b.keys == p
    p, off = _obtain(off, 4 * R); assert b.values_unsorted == p
    p, off = _obtain(off, 4 * R); assert b.values == p
    p, off = _obtain(off, b.sorting_size); assert b.sorting_space == p
    assert end == off == L.gsr_required_binning(R)


def test_higher_msb_matches_the_reference_values():
    L = _capi.lib()
    for tiles, want in ((64, 7), (3072, 12), (8160, 13), (32400, 15)):
        assert L.gsr_higher_msb(tiles) == want == cpu_oracle.higher_msb(tiles)
    for n in list(range(1, 70000, 37)):
        assert L.gsr_higher_msb(n) == cpu_oracle.higher_msb(n)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _capi.lib()


def _compile_and_run(tmp_path, source: str, flags=(), cxx=False):
    import subprocess
    src = tmp_path / ("probe.cpp" if cxx else "probe.c")
    src.write_text(source)
    exe = tmp_path / "probe"
    cmd = (["g++", "-std=c++17"] if cxx else ["gcc", "-std=c11"]) + ["-I", os.path.join(ROOT, "include"), *flags, str(src), "-o", str(exe)]
    subprocess.check_call(cmd)
    return subprocess.check_output([str(exe)], text=True)


def test_ctypes_structs_have_the_headers_layout(tmp_path):
    """The header is compiled as C and asked for sizes / offsets; the ctypes mirrors must agree field for field (a
    silent mismatch would shift every pointer behind it)."""
    probes = {"gsr_forward_args": (_capi.ForwardArgs, ["flags", "num_gaussians", "out_color", "stream", "tile_history", "num_rendered",
                                                        "records_staged", "stage_ms", "plan_used", "receipt"]),
              "gsr_backward_args": (_capi.BackwardArgs, ["point_list", "dL_dout_color", "dL_drotations", "stage_ms", "sh_dims",
                                                          "receipt"]),
              "gsr_forward_receipt": (_capi.ForwardReceipt, ["plan_used", "tile_row_end", "num_visible", "serial",
                                                             "geometry_chunk", "binning_chunk", "async_words", "tile_history"])}
    lines = []
    for cname, (_, fields) in probes.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    out = _compile_and_run(tmp_path, '#include <stdio.h>\n#include <stddef.h>\n#include "gsrast_amd.h"\nint main(void){' + "".join(lines) + "return 0;}")
    got = dict(line.split() for line in out.strip().splitlines())
    for cname, (ctype, fields) in probes.items():
        assert int(got[cname]) == C.sizeof(ctype), cname
        for f in fields:
            assert int(got[f"{cname}.{f}"]) == getattr(ctype, f).offset, f"{cname}.{f}"


def test_python_constants_are_the_headers(tmp_path):
    """Every GSR_FLAG_* / GSR_PLAN_* / GSR_ERR_* the Python mirror names has the header's value (the header compiled as C prints
    them): a flag added on one side only would silently select something else."""
    names = sorted(k for k in vars(_capi) if k.startswith(("GSR_FLAG_", "GSR_PLAN_", "GSR_ERR_")) or k == "GSR_OK")
    assert {"GSR_FLAG_NO_TILE_HISTORY", "GSR_FLAG_SERIAL_EMIT", "GSR_PLAN_EMIT_OVERLAPPED", "GSR_PLAN_COLORS_BESIDE",
            "GSR_PLAN_TILE_ORDER_DROPPED"} <= set(names)
    body = "".join(f'printf("{k} %lld\\n", (long long)({k}));' for k in names)
    out = _compile_and_run(tmp_path, '#include <stdio.h>\n#include "gsrast_amd.h"\nint main(void){' + body + "return 0;}")
    got = dict(line.split() for line in out.strip().splitlines())
    for k in names:
        assert int(got[k]) == int(getattr(_capi, k)), k


def test_poll_async_error_wants_a_receipt():
    L = _capi.lib()
    assert L.gsr_poll_async_error(None) == _capi.GSR_ERR_INVALID_ARG
    assert L.gsr_poll_async_error(C.byref(_capi.ForwardReceipt())) == _capi.GSR_ERR_INVALID_ARG
    # a kernel that gives up writes its call's serial: words {N-sized sort, R-sized sort, owner, 0}
    words = (C.c_uint32 * 4)(7, 0, 7, 0)
    r = _capi.ForwardReceipt()
    r.magic, r.serial, r.async_words = _capi.GSR_RECEIPT_MAGIC, 6, C.addressof(words)
    assert L.gsr_poll_async_error(C.byref(r)) == _capi.GSR_ERR_STALE_RECEIPT      # the slot has a new owner: unknown, said so
    r.serial = 7
    assert L.gsr_poll_async_error(C.byref(r)) == _capi.GSR_ERR_INTERNAL
    words[0] = 6                                                                  # a late writer of the slot's PREVIOUS owner
    assert L.gsr_poll_async_error(C.byref(r)) == _capi.GSR_OK                     # ... does not raise the new owner's flag
    r.serial = 6
    assert L.gsr_poll_async_error(C.byref(r)) == _capi.GSR_ERR_INTERNAL           # ... and is still reported to whoever holds its receipt
    words[0], words[1] = 0, 7
    r.serial = 7
    assert L.gsr_poll_async_error(C.byref(r)) == _capi.GSR_ERR_INTERNAL
    words[1] = 0
    assert L.gsr_poll_async_error(C.byref(r)) == _capi.GSR_OK


@pytest.mark.parametrize("with_glm", [False, True])
def test_shim_state_pointers_are_glm_typed_when_glm_exists(tmp_path, with_glm):
    """AuxBuffer.cuh:46-49,57 declare glm::vec2* / vec4* / vec3* / uvec2*: with glm on the include path the shim's structs
    carry exactly those types (a caller's `glm::vec2* p = geomState.means2D;` compiles), without it PODs of the same layout.
    This image has no glm; tests/doubles/glm is a stand-in with glm's default layout, for this compile test only."""
    flags = ["-I", os.path.join(ROOT, "tests", "doubles")] if with_glm else []
    body = """
#include <cstdio>
#include <type_traits>
#include "gscuda_shim.hpp"
int main() {
    char* chunk = nullptr;
    gscuda::gs::GeometryState g = gscuda::gs::GeometryState::fromChunk(chunk, 10);
    chunk = nullptr;
    gscuda::gs::ImageState im = gscuda::gs::ImageState::fromChunk(chunk, 64);
#ifdef GSCUDA_SHIM_HAS_GLM
    glm::vec2* m = g.means2D; glm::vec4* c = g.conicOpacity; glm::vec3* rgb = g.rgb; glm::uvec2* r = im.ranges;
    static_assert(std::is_same<gscuda::vec3, glm::vec3>::value, "alias");
    std::printf("glm %d\\n", (int)(m != nullptr && c != nullptr && rgb != nullptr) + (int)(r == nullptr));
#else
    gscuda::vec2* m = g.means2D; (void)m; (void)im;
    std::printf("pod 0\\n");
#endif
    return gscuda::required<gscuda::gs::GeometryState>(10) == gsr_required_geometry(10) ? 0 : 1;
}
"""
    import subprocess
    src = tmp_path / "shim_probe.cpp"
    src.write_text(body)
    exe = tmp_path / "shim_probe"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), *flags, str(src), "-o", str(exe),
                           _capi.LIB_PATH, "-Wl,-rpath," + os.path.dirname(_capi.LIB_PATH), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True)
    assert out.split()[0] == ("glm" if with_glm else "pod")


def test_device_shape_follows_the_compute_unit_count():
    """What the launch heuristics assume about the chip comes from its CU count (hipDeviceAttributeMultiprocessorCount), not
    from constants: a whole MI355X (256 CUs) and one eighth of it (32 CUs, a CPX partition)."""
    L = _capi.lib()
    out = (C.c_uint32 * 4)()
    L.gsr_device_shape(256, out)
    assert list(out) == [256, 5120, 3072, 5120]
    L.gsr_device_shape(32, out)
    assert list(out) == [32, 640, 384, 640]
    L.gsr_device_shape(0, out)                      # (never a division by zero downstream)
    assert out[0] == 1 and out[1] == 20


def test_thread_release_without_any_call_is_a_no_op():
    L = _capi.lib()
    assert L.gsr_thread_release() == _capi.GSR_OK and L.gsr_thread_release() == _capi.GSR_OK
