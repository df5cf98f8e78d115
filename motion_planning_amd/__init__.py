"""MI355X-native lattice motion-planning core.

SL path DP, ST speed DP and Cartesian->Frenet conversion of the
ZJUDriving/motion_planning planner as hand-written gfx950 HIP kernels behind a
C ABI (include/mplattice.h), with the reference's Python classes kept as the
front end (motion_planning_amd.Planning / .Model) and batched entry points in
motion_planning_amd.batch.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
