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
