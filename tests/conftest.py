import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture
def knob():
    """Sets testing knobs of the library for one test (mp_set_knob; the library reads the
    environment only once, so tests cannot switch code paths through os.environ) and puts
    them back to "not set" afterwards."""
    from motion_planning_amd import _lib
    used = []

    def set_(name, value):
        _lib.set_knob(name, value)
        used.append(name)

    yield set_
    for name in used:
        _lib.set_knob(name, -1)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
