"""Robot-frame scene container read by SLMap (interface of the reference's
Planning/Interface/robot_map.py and Model/general_converter.py; no CARLA)."""
import numpy as np

from ...Model.obstacle import DynamicObstacle, StaticObstacle, VehicleBox
from ...Utils.tool import cal_dist, check_static, to_point


class GeneralConverter:
    """Rigid 2-D transform world <-> robot: p_w = R p_r + t."""

    def __init__(self, R, t):
        self.R = np.asarray(R, dtype=np.float64)
        self.t = np.asarray(t, dtype=np.float64)

    def world_to_robot(self, point):
        return self.R.T @ (np.asarray(point, dtype=np.float64) - self.t)

    def robot_to_world(self, point):
        return self.R @ np.asarray(point, dtype=np.float64) + self.t

    def vel_world_to_robot(self, vel):
        return self.R.T @ np.asarray(vel, dtype=np.float64)


class RobotMap:
    def __init__(self, R, t):
        self.WRC = GeneralConverter(R, t)
        self.ori_point = np.array([0.0, 0.0])
        self.ignore_dist = 20  # obstacles farther than this never enter the map
        self.static_ob = []
        self.dynamic_ob = []
        self.robot = []

    def add_robot(self, rob_pos, rob_vel, rob_dist):
        self.robot = VehicleBox(self.WRC.world_to_robot(rob_pos), self.WRC.vel_world_to_robot(rob_vel), rob_dist)

    def add_ref_line(self, ref_line, l_width, n_l, n_s, cal_theta_ind=0):
        self.n_l, self.n_s, self.l_width = n_l, n_s, l_width
        self.save_width = 0.6
        pts = [self.WRC.world_to_robot(p) for p in ref_line]
        self.rx_list = [p[0] for p in pts]
        self.ry_list = [p[1] for p in pts]
        self.line_vec = 0.0

    def add_obstacle(self, ob_pos, ob_vel, ob_dist):
        """Keeps an obstacle unless it is more than 1 m behind the robot or (at its
        closest predicted position) beyond ignore_dist (robot_map.py:47-68)."""
        ob_pos = self.WRC.world_to_robot(ob_pos)
        if ob_pos[0] < -1:
            return False
        if check_static(ob_vel):
            if cal_dist(self.ori_point, ob_pos) < self.ignore_dist:
                self.static_ob.append(StaticObstacle(ob_pos, ob_vel, ob_dist))
                return True
            return False
        dy_ob = DynamicObstacle(ob_pos, self.WRC.vel_world_to_robot(ob_vel), ob_dist)
        if dy_ob.cal_min_dist(self.ori_point) < self.ignore_dist:
            self.dynamic_ob.append(dy_ob)
            return True
        return False

    def path_convert(self, point_list):
        return [self.WRC.robot_to_world(to_point(p[0], p[1])) for p in point_list]
