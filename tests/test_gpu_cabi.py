"""The C ABI from a plain C++ program (HIP runtime only, no Python or torch in the
process): tests/cabi/cabi_check.cpp, built by __graft_entry__.build()."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cabi_from_cpp():
    exe = os.path.join(HERE, "cabi", "cabi_check")
    assert os.path.exists(exe), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi ok" in r.stdout, r.stdout
