"""SL (Frenet) lattice map: interface of the reference's Planning/DP_Path/sl_map.py.

The Frenet frame (reference-line tables) and the conversion of every obstacle
position -- each static obstacle and all 11 predicted frames of each dynamic
one -- run on the GPU in one batched launch.
"""
import numpy as np

from ...Model.cartesian_frenet_conversion import CartesianFrenetConverter
from ...Model.obstacle import ObstacleTrajectory, StaticObstacle
from ...Utils.tool import get_arange, to_point


class SLMap:
    def __init__(self, robot_map):
        self.RMP = robot_map
        self.n_l = robot_map.n_l
        self.n_s = robot_map.n_s  # replaced by the number of generated layers
        self.mid_l = int((self.n_l - 1) / 2)
        self.l_width = robot_map.l_width
        self.save_width = robot_map.save_width
        self.ob_list = []    # Frenet centres of everything that can touch the lattice
        self.ob_dist = 0.0   # one collision radius: the largest among them
        self.d_s = 8         # layer spacing along the reference line
        self.st_ob_traj = []
        self.dy_ob_traj = []
        self.create_frenet_map()
        self.add_obstacle()

    def create_frenet_map(self):
        rmp = self.RMP
        self.converter = CartesianFrenetConverter(0.0, 0.0, np.array(rmp.rx_list), np.array(rmp.ry_list),
                                                  rmp.line_vec)
        vel = rmp.robot.vel
        so, lo, s_dt = self.converter.cartesian_to_frenet(0.0, 0.0, vel[0], vel[1], order=1)
        self.ego_point = np.array([so, lo])
        self.robot_vel = s_dt
        end_s = self.converter.get_s(rmp.rx_list[-1])
        if end_s > 100:
            print("Error, frenet map too big")
            raise ValueError
        # layers every d_s; the last one sits on the end of the line and absorbs
        # a remainder of up to 1.5 d_s (sl_map.py:61-66)
        s_list = []
        s = self.d_s
        while True:
            s_list.append(s)
            s += self.d_s
            if s + self.d_s / 2 >= end_s:
                s_list.append(end_s)
                break
        self.n_s = len(s_list)
        half = (self.l_width - self.save_width) / 2.0
        l_row = np.linspace(0.0 - half, 0.0 + half, num=self.n_l)
        self.s_map = np.repeat(np.array(s_list, dtype=np.float64)[:, None], self.n_l, axis=1)
        self.l_map = np.repeat(l_row[None, :], self.n_s, axis=0)

    def check_in_line(self, point, radius):
        return abs(point[1]) - radius < self.l_width

    def add_obstacle(self):
        """Frenet positions of all obstacle frames in one GPU launch, then the
        reference's bookkeeping (sl_map.py:86-109)."""
        statics, dynamics = self.RMP.static_ob, self.RMP.dynamic_ob
        pts = [ob.ob_pos for ob in statics] + [p for ob in dynamics for p in ob.ob_pos_list]
        if pts:
            pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
            s_all, l_all = self.converter.cartesian_to_frenet_batch(pts[:, 0], pts[:, 1])
        k = 0
        for ob in statics:
            point = np.array([s_all[k], l_all[k]])
            k += 1
            self.st_ob_traj.append(StaticObstacle(point, 0.0, ob.ob_dist))
            if self.check_in_line(point, ob.ob_dist):
                self.ob_list.append(point)
                self.ob_dist = max(self.ob_dist, ob.ob_dist)
        for ob in dynamics:
            traj = ObstacleTrajectory(ob.ob_dist)
            for _ in ob.ob_pos_list:
                point = np.array([s_all[k], l_all[k]])
                k += 1
                traj.ob_pos_list.append(point)
                if self.check_in_line(point, ob.ob_dist):
                    self.ob_list.append(point)
                    self.ob_dist = max(self.ob_dist, ob.ob_dist)
            self.dy_ob_traj.append(traj)

    def path_sampling(self, curve_path, draw=False, ss=[]):
        """(s, l) samples of a path curve every curve.dt, or at the given ss (sl_map.py:114-126)."""
        if len(ss) == 0:
            s_st = curve_path.t_bios
            ss = get_arange(s_st, s_st + np.sum(curve_path.T), curve_path.dt)
        ll = curve_path.calc_point_arr(ss, 0)
        return [to_point(ss[j], ll[j]) for j in range(len(ss))]

    def path_convert(self, point_list):
        """Frenet -> robot frame (sl_map.py:129-134)."""
        if len(point_list) == 0:
            return []
        pts = np.asarray(point_list, dtype=np.float64).reshape(-1, 2)
        xs, ys = self.converter.frenet_to_cartesian_batch(pts[:, 0], pts[:, 1])
        return [to_point(x, y) for x, y in zip(xs, ys)]
