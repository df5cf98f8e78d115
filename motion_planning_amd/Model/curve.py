"""Quintic polynomial and piecewise-quintic spline on the host.

Same interface and evaluation rules as the reference's Model/curve.py; these
objects carry planner OUTPUTS (the chosen path l(s), the speed profile s(t)),
a handful of knots each, so they stay in NumPy.  The reference-line side of the
same mathematics (hundreds of samples per scenario, thousands of scenarios)
runs on the GPU: see batch.refline_tables.
"""
import numpy as np

from ..Utils.tool import cal_dist_arr, get_arange


class QuinticPoly:
    """x(t) on [0, T] matching position, velocity and acceleration at both ends
    (reference Model/curve.py:129-186)."""

    def __init__(self, xs, vxs, axs, xe, vxe, axe, T):
        self.xs, self.vxs, self.axs = xs, vxs, axs
        self.xe, self.vxe, self.axe = xe, vxe, axe
        self.a0 = xs
        self.a1 = vxs
        self.a2 = axs / 2.0
        # closed-form solution of the 3x3 end-condition system (curve.py:155-157)
        self.a3 = 1.0 * (20 * xe - 20 * xs - (8 * vxe + 12 * vxs) * T - (3 * axs - axe) * T * T) / (2 * T * T * T)
        self.a4 = 1.0 * (30 * xs - 30 * xe + (14 * vxe + 16 * vxs) * T + (3 * axs - 2 * axe) * T * T) / (2 * T * T * T * T)
        self.a5 = 1.0 * (12 * xe - 12 * xs - (6 * vxe + 6 * vxs) * T - (axs - axe) * T * T) / (2 * T * T * T * T * T)

    def calc_point(self, t, order):
        """Value (order 0) or derivative (1..3) at t, power form (curve.py:159-173)."""
        a0, a1, a2, a3, a4, a5 = self.a0, self.a1, self.a2, self.a3, self.a4, self.a5
        if order == 0:
            return a0 + a1 * t + a2 * t ** 2 + a3 * t ** 3 + a4 * t ** 4 + a5 * t ** 5
        if order == 1:
            return a1 + 2 * a2 * t + 3 * a3 * t ** 2 + 4 * a4 * t ** 3 + 5 * a5 * t ** 4
        if order == 2:
            return 2 * a2 + 6 * a3 * t + 12 * a4 * t ** 2 + 20 * a5 * t ** 3
        if order == 3:
            return 6 * a3 + 24 * a4 * t + 60 * a5 * t ** 2
        print("Error, order is not valid")
        return 0.0

    def calc_arc_len(self, t0, t1, dt):
        """Rectangle-rule arc length over the end-inclusive sample set (curve.py:175-186)."""
        sign, lo, hi = (1, t0, t1) if t0 <= t1 else (-1, t1, t0)
        t = get_arange(lo, hi, dt)
        if len(t) == 0:
            return 0.0
        slope = self.calc_point(t, 1)
        return sign * dt * np.sum(np.sqrt(1 + slope ** 2))


class Curve:
    """Piecewise quintic through knots (t_i, x_i) (reference Model/curve.py:13-126).

    Knot slopes are central differences (end slopes = vec), knot accelerations
    central differences of the slopes (ends 0)."""

    def __init__(self, t, dt, x, vec):
        t = np.asarray(t, dtype=np.float64)
        x = np.asarray(x, dtype=np.float64)
        self.T = np.diff(t)
        self.n = len(self.T)
        self.dt = dt
        self.t_bios = t[0]
        self.t_end = t[-1]
        span = self.T[1:] + self.T[:-1]
        v = np.concatenate(([vec], 1.0 * (x[2:] - x[:-2]) / span, [vec]))
        a = np.concatenate(([0.0], 1.0 * (v[2:] - v[:-2]) / span, [0.0]))
        self.curve_list = [QuinticPoly(x[i], v[i], a[i], x[i + 1], v[i + 1], a[i + 1], self.T[i])
                           for i in range(self.n)]

    def get_i(self, t):
        """Segment of local time t and the time inside it (curve.py:80-91); past
        the end: the last segment and the TOTAL length."""
        if t < 0.0:
            print("Error, t is smaller than 0!")
            raise IndexError
        t_sum = 0.0
        for i in range(self.n):
            t_sum += self.T[i]
            if t <= t_sum:
                return i, t - t_sum + self.T[i]
        return self.n - 1, t_sum

    def calc_point(self, t, order):
        i, tt = self.get_i(t - self.t_bios)
        return self.curve_list[i].calc_point(tt, order)

    def calc_point_arr(self, t_arr, order):
        """Evaluation on an increasing sample array (curve.py:40-73).  The segment
        advances only once a sample lies more than dt beyond the segment end (and
        by one segment per sample), so samples just after a knot still use the
        previous segment -- unlike calc_point."""
        local = np.asarray(t_arr, dtype=np.float64) - self.t_bios
        seg, _ = self.get_i(local[0])
        t_sum = np.sum(self.T[:seg + 1])
        out = np.empty(len(local))
        start = 0
        for j in range(len(local)):
            if local[j] > t_sum + self.dt:
                out[start:j] = self.curve_list[seg].calc_point(local[start:j] - t_sum + self.T[seg], order)
                seg += 1
                start = j
                if seg >= self.n:
                    print("Error, out of limit")
                    raise IndexError
                t_sum += self.T[seg]
        out[start:] = self.curve_list[seg].calc_point(local[start:] - t_sum + self.T[seg], order)
        return out

    def calc_arc_len(self, t0, t1):
        """Arc length between two abscissae, one rectangle-rule piece per segment
        (curve.py:93-115); 0 when t1 lies before t0."""
        t0 = t0 - self.t_bios
        t1 = t1 - self.t_bios
        if t0 > t1:
            return 0.0
        st_i, st_t = self.get_i(t0)
        en_i, en_t = self.get_i(t1)
        if en_t > self.T[en_i]:
            en_t = self.T[en_i]
            print("Error, out of T limit")
        if st_i == en_i:
            return self.curve_list[st_i].calc_arc_len(st_t, en_t, self.dt)
        total = self.curve_list[st_i].calc_arc_len(st_t, self.T[st_i], self.dt)
        for i in range(st_i + 1, en_i):
            total += self.curve_list[i].calc_arc_len(0, self.T[i], self.dt)
        total += self.curve_list[en_i].calc_arc_len(0, en_t, self.dt)
        return total

    def projection(self, t, x):
        """Nearest of the curve's dt-spaced samples to (t, x), no refinement
        (curve.py:117-126): returns (distance, sample point)."""
        t_arr = get_arange(0, np.sum(self.T), self.dt) + self.t_bios
        x_arr = self.calc_point_arr(t_arr, 0)
        return cal_dist_arr(np.vstack((t_arr, x_arr)), np.array([t, x]))
