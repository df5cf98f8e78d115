"""ST (time x arc length) grid: interface of the reference's Planning/DP_Speed/st_map.py.
add_obstacle() finds, on the GPU, the arc-length interval of the path that every
obstacle blocks at every time layer."""
import math

import numpy as np

from ... import _lib, batch as mb


class ObState:
    """Blocked intervals [s_down[k], s_up[k]] at one time layer."""

    def __init__(self, t):
        self.t = t
        self.s_down = []
        self.s_up = []


class PointState:
    def __init__(self, t, s):
        self.t = t
        self.s = s
        self.cost = 0.0
        self.pre_ind = -1


class STMap:
    def __init__(self, converter, s_end, t_end=5):
        self.converter = converter
        self.ds = 1.0
        self.dt = 0.5
        self.n_s = int(math.floor(1.0 * s_end / self.ds)) + 1
        self.n_t = int(math.ceil(1.0 * t_end / self.dt)) + 1
        self.s_end = self.n_s * self.ds
        self.t_end = self.n_t * self.dt
        self.map = [[PointState(i * self.dt, j * self.ds) for j in range(self.n_s)] for i in range(self.n_t)]
        self.ob_mat = []

    def add_obstacle(self, path, st_ob_traj, dy_ob_traj):
        """One ObState per time layer: statics first, then frame i of each dynamic
        obstacle; an interval spans the first and last path sample closer (in
        Frenet space) than the obstacle radius (st_map.py:24-60)."""
        path = np.asarray(path, dtype=np.float64).reshape(-1, 2)
        n_t = self.n_t
        st = np.array([[o.ob_pos[0], o.ob_pos[1], o.ob_dist] for o in st_ob_traj], dtype=np.float64).reshape(-1, 3)
        n_dy = len(dy_ob_traj)
        dy = np.zeros((max(n_dy, 1), n_t, 2))
        for q, o in enumerate(dy_ob_traj):
            dy[q] = np.asarray(o.ob_pos_list, dtype=np.float64)[:n_t]  # IndexError if a frame is missing
        dy_r = np.array([o.ob_dist for o in dy_ob_traj] or [0.0])
        r_max = max(1, len(st) + n_dy)
        down, up, n_reg, status = mb.st_regions(
            n_t, path[None, :, 0], path[None, :, 1], np.array([len(path)], np.int32),
            st[None, :, 0], st[None, :, 1], st[None, :, 2], np.array([len(st)], np.int32),
            dy[None, :, :, 0], dy[None, :, :, 1], dy_r[None], np.array([n_dy], np.int32), r_max)
        assert int(status[0]) == _lib.MP_ST_OK
        down, up, n_reg = down[0].cpu().numpy(), up[0].cpu().numpy(), n_reg[0].cpu().numpy()
        for i in range(n_t):
            state = ObState(i * self.dt)
            state.s_down = [float(v) for v in down[i, :n_reg[i]]]
            state.s_up = [float(v) for v in up[i, :n_reg[i]]]
            self.ob_mat.append(state)
