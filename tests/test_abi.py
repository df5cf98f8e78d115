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
    assert L.mp_version() == 110


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


def test_missing_library_fails_loudly(monkeypatch):
    from motion_planning_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmplattice.so")
    with pytest.raises(ImportError):
        _lib.lib()
