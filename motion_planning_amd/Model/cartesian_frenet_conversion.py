"""Cartesian <-> Frenet conversion about a spline reference line y(x).

Interface of the reference's Model/cartesian_frenet_conversion.py.  The
constructor and cartesian_to_frenet run on the GPU (reference-line tables +
nearest-sample projection, batch.refline_tables / batch.c2f); one converter is
a batch of one scenario; cartesian_to_frenet_batch / frenet_to_cartesian_batch
convert many points in one launch.
"""
import numpy as np

from .. import _lib, batch as mb
from ..Utils.tool import get_arange
from .curve import Curve


class CartesianFrenetConverter:
    def __init__(self, ex, ey, rx_list, ry_list, vec):
        self.d_x = 0.5  # sample spacing of the reference line (cfc.py:12)
        rx = np.asarray(rx_list, dtype=np.float64)
        ry = np.asarray(ry_list, dtype=np.float64)
        self.ref_curve = Curve(rx, self.d_x, ry, vec)
        self.tables = mb.refline_tables(rx[None], ry[None], np.array([len(rx)], np.int32),
                                        np.array([float(vec)]), np.array([[float(ex), float(ey)]]), self.d_x)
        st = int(self.tables.status[0])
        if st == _lib.MP_RL_E_INDEX:
            raise IndexError("reference line too short to sample")
        if st != _lib.MP_RL_OK:
            raise ValueError("reference line exceeds the kernel's capacity (status %d)" % st)
        ori = self.tables.origin[0].cpu().numpy()
        self.rx_ori = float(ori[0])  # Frenet origin = projection of (ex, ey)
        self.ry_ori = float(ori[1])
        self.rx_max = rx_list[-1]

    def cartesian_to_frenet_batch(self, xs, ys, vxs=None, vys=None):
        """Vector form: arrays in, arrays (s, l[, s_dt]) out."""
        xs = np.atleast_1d(np.asarray(xs, dtype=np.float64))
        scn = np.zeros(len(xs), np.int32)
        out = mb.c2f(self.tables, xs, ys, scn, vxs, vys)
        return tuple(o.cpu().numpy() for o in out)

    def cartesian_to_frenet(self, x, y, vx=0.0, vy=0.0, order=0):
        """(s, l) of a point, plus s_dt for order 1 (cfc.py:32-53)."""
        if order == 0:
            s, l = self.cartesian_to_frenet_batch([x], [y])
            return float(s[0]), float(l[0])
        s, l, s_dt = self.cartesian_to_frenet_batch([x], [y], [vx], [vy])
        return float(s[0]), float(l[0]), float(s_dt[0])

    def get_s(self, rx):
        return self.ref_curve.calc_arc_len(self.rx_ori, rx)

    def get_curvature(self, rx):
        dl = self.ref_curve.calc_point(rx, 1)
        ddl = self.ref_curve.calc_point(rx, 2)
        return 1.0 * abs(ddl) / ((1 + dl ** 2) ** 1.5)

    def get_rxy(self, s):
        """First 0.5 m abscissa from the origin whose arc length reaches s; the
        end of the line if none does (cfc.py:72-86)."""
        rx = None
        for cand in get_arange(self.rx_ori, self.rx_max, self.d_x):
            if s <= self.ref_curve.calc_arc_len(self.rx_ori, cand):
                rx = cand
                break
        if rx is None:
            rx = self.rx_max
            print("Error, cannot get rxy")
        return rx, self.ref_curve.calc_point(rx, 0)

    def frenet_to_cartesian_batch(self, ss, ls):
        """Vector form of frenet_to_cartesian: one GPU launch over the
        Frenet->Cartesian table built with the converter."""
        ss = np.atleast_1d(np.asarray(ss, dtype=np.float64))
        x, y = mb.f2c(self.tables, ss, ls, np.zeros(len(ss), np.int32))
        return x.cpu().numpy(), y.cpu().numpy()

    def frenet_to_cartesian(self, s, l):
        """(x, y) of Frenet point (s, l): the first 0.5 m abscissa whose arc length
        reaches s, offset by l along the normal there (cfc.py:55-60)."""
        x, y = self.frenet_to_cartesian_batch([s], [l])
        return float(x[0]), float(y[0])
