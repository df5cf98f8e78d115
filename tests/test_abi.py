"""The C-ABI shared object loads and exports every symbol include/mplattice.h
declares (no compute calls: runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mplattice.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from motion_planning_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 9
    for s in syms:
        assert hasattr(L, s), s
        assert s in _lib.SIGNATURES, "no ctypes signature for " + s
    assert set(_lib.SIGNATURES) == set(syms)
    assert L.mp_version() == _lib.MP_VERSION
    hdr = open(os.path.join(ROOT, "include", "mplattice.h")).read()
    assert int(re.search(r"#define MP_VERSION (\d+)", hdr).group(1)) == _lib.MP_VERSION


def test_host_side_argument_errors_need_no_gpu():
    import ctypes as C
    from motion_planning_amd import _lib
    L = _lib.lib()
    assert L.mp_sl_plan(None, None, None, None, None, None, None, None, None) == -1
    assert b"null" in L.mp_last_error()
    prm = _lib.SlParams(0.5, 0.5, 1e4, 1, 2)
    assert L.mp_sl_dp(0, 1, 1, None, None, C.byref(prm), None, None, None, None, None, None, None) == -1
    assert L.mp_st_dp(1, 1, 5, None, None, None, None, 0, None, 1.0, 1.0, 0.5, None, None, None, None, None) == -1
    assert L.mp_c2f_batch(0, *([None] * 5), 4, *([None] * 10)) == -1


def test_plan_arena_layout_needs_no_gpu():
    """The arena layout of mp_plan_scenes: arrays are 256-byte aligned, do not overlap
    and fit in the reported size; optional tables take no space unless asked for."""
    import ctypes as C
    from motion_planning_amd import _lib
    L = _lib.lib()
    for want_maps in (0, 1):
        dims = _lib.PlanDims(7, 22, 5, 3, 11, 5, 232, 14, 208, 11, 102, 8, 54, want_maps)
        total = L.mp_plan_arena_bytes(C.byref(dims))
        assert total > 0
        name, dtype, off = C.c_char_p(), C.c_int32(), C.c_int64()
        shape = (C.c_int64 * 4)()
        spans, names = [], []
        for i in range(L.mp_plan_array_count()):
            assert L.mp_plan_array_info(C.byref(dims), i, C.byref(name), C.byref(dtype), shape, C.byref(off)) == 0
            n = shape[0] * shape[1] * shape[2] * shape[3] * (8 if dtype.value == _lib.MP_F64 else 4)
            assert off.value % 256 == 0 and off.value + n <= total
            spans.append((off.value, off.value + n))
            names.append(name.value.decode())
            if names[-1] in ("cost_map", "index_map", "st_cost", "st_pre"):
                assert (n > 0) == bool(want_maps)
        assert len(set(names)) == len(names) and {"status", "final_x", "final_y", "vv", "path_idx"} <= set(names)
        spans.sort()
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    bad = _lib.PlanDims(0, 22, 5, 3, 11, 5, 232, 14, 208, 11, 102, 8, 54, 0)
    assert L.mp_plan_arena_bytes(C.byref(bad)) == -1
    prm = _lib.PlanParams()
    L.mp_plan_default_params(C.byref(prm))
    assert (prm.l_width, prm.layer_gap, prm.sl.d_s, prm.st_dt) == (3.5, 8.0, 0.5, 0.5)
    assert L.mp_plan_scenes(None, C.byref(dims), C.byref(prm), None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    from motion_planning_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmplattice.so")
    with pytest.raises(ImportError):
        _lib.lib()
