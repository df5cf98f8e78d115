"""Centre-lane follower with the planner contract (reference
Planning/DP_Path/track_path_planner.py): no search, always the guide lane."""
import numpy as np

from ...Model.curve import Curve


class TrackPathPlanner:
    def __init__(self, sl_map):
        self.sl_map = sl_map
        self.d_s = 0.5

    def plan(self):
        m = self.sl_map
        path_s = np.concatenate(([m.ego_point[0]], np.asarray(m.s_map)[:, m.mid_l]))
        path_l = np.concatenate(([m.ego_point[1]], np.asarray(m.l_map)[:, m.mid_l]))
        return True, Curve(path_s, self.d_s, path_l, 0.0)
