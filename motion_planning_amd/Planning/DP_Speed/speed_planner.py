"""ST lattice speed planner: interface of the reference's Planning/DP_Speed/speed_planner.py.
find_speed() runs the whole dynamic programme in one GPU launch (a batch of one
scenario) and writes cost / pre_ind back into st_map.map as the reference does."""
import numpy as np

from ... import batch as mb
from ...Model.curve import Curve
from ...Utils.tool import get_arange


class SpeedPlanner:
    def __init__(self, st_map, cur_vel, speed_lim):
        self.st_map = st_map
        self.cur_vel = cur_vel
        self.speed_lim = speed_lim
        self.dt = st_map.dt
        self.ds = st_map.ds
        self.curve_dt = self.dt * 1.0 / 5   # 0.1 s output spacing
        self.path_s_ind = []

    def find_speed(self):
        m = self.st_map
        n_t, n_s = m.n_t, m.n_s
        r_max = max([1] + [len(o.s_down) for o in m.ob_mat])
        down = np.zeros((1, n_t, r_max))
        up = np.zeros((1, n_t, r_max))
        n_reg = np.zeros((1, n_t), np.int32)
        for i, o in enumerate(m.ob_mat[:n_t]):
            n_reg[0, i] = len(o.s_down)
            down[0, i, :n_reg[0, i]] = o.s_down
            up[0, i, :n_reg[0, i]] = o.s_up
        out = mb.st_dp(np.array([n_s], np.int32), down, up, n_reg, np.array([float(self.cur_vel)]),
                       self.speed_lim, self.ds, self.dt)
        cost = out["cost"][0].cpu().numpy()
        pre = out["pre"][0].cpu().numpy()
        for i in range(1, n_t):
            row = m.map[i]
            for j in range(n_s):
                row[j].cost = float(cost[i, j])
                row[j].pre_ind = int(pre[i, j])
        plen = int(out["path_len"][0])
        self.path_s_ind = [int(j) for j in out["path_s_ind"][0, :plen].cpu().numpy()]
        return True

    def plan(self):
        """(ss, vv): arc length and speed every 0.1 s along the quintic spline
        through the chosen ST nodes (speed_planner.py:25-55)."""
        self.find_speed()
        s_arr = np.array([j * self.ds for j in self.path_s_ind])
        t_arr, t = [], 0.0
        for _ in self.path_s_ind:
            t_arr.append(t)
            t += self.dt
        t_arr = np.array(t_arr)
        curve_speed = Curve(t_arr, self.curve_dt, s_arr, self.cur_vel)
        tt = get_arange(t_arr[0], t_arr[-1], self.curve_dt)
        return curve_speed.calc_point_arr(tt, 0), curve_speed.calc_point_arr(tt, 1)
