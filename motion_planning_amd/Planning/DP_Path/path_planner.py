"""SL lattice path planner: interface of the reference's Planning/DP_Path/path_planner.py.

find_path() runs the whole dynamic programme (edge costs, layer-wise
min-relaxation, abort test, backtrack) in one GPU launch; one planner is a
batch of one scenario.  There is no CPU fallback.
"""
import numpy as np

from ... import _lib, batch as mb
from ...Model.curve import Curve, QuinticPoly
from ...Utils.tool import get_arange


class PathPlanner:
    def __init__(self, sl_map):
        self.sl_map = sl_map
        self.init_ind = -2
        self.cost_map = np.zeros((sl_map.n_s, sl_map.n_l))
        self.index_map = self.init_ind * np.ones((sl_map.n_s, sl_map.n_l))
        self.d_s = 0.5             # sample spacing along an edge
        self.w_d = 0.5             # guide / smoothness mix
        self.path_ind_list = []
        self.no_path_cost = 1e4    # obstacle penalty 3x, abort threshold 10x
        self._edge_key = self._edge = None

    def _inputs(self):
        m = self.sl_map
        ob = np.asarray(m.ob_list, dtype=np.float64).reshape(-1, 2)
        return (np.asarray(m.s_map, dtype=np.float64), np.asarray(m.l_map, dtype=np.float64),
                np.asarray(m.ego_point, dtype=np.float64), ob, float(m.ob_dist))

    def _batch(self):
        """The lattice as a batch of one for the fused kernels, which need every layer at a
        single arc length (SLMap builds it so, sl_map.py:69-78); None for any other lattice."""
        s_map, l_map, ego, ob, ob_dist = self._inputs()
        if not (s_map == s_map[:, :1]).all():
            return None
        return mb.SLBatch(s_map[None, :, 0], l_map[None], ego[None], ob[None, :, 0], ob[None, :, 1],
                          np.array([len(ob)], np.int32), np.array([ob_dist]))

    def _params(self):
        return mb.sl_params(self.d_s, self.w_d, self.no_path_cost, True, self.sl_map.mid_l)

    def _edge_tensor(self, sb=None, general=False):
        """(edge costs [1, n_s, n_l, n_l], status) of the lattice as it stands, kept until the
        lattice, the obstacles or the constants change.  The layer kernel serves lattices with
        one arc length per layer and up to 128 samples per edge; anything else goes through
        the per-node kernel."""
        s_map, l_map, ego, ob, ob_dist = self._inputs()
        key = (s_map.tobytes(), l_map.tobytes(), ego.tobytes(), ob.tobytes(), ob_dist, self.d_s, self.w_d,
               self.no_path_cost, self.sl_map.mid_l)
        if self._edge_key == key:
            return self._edge
        sb = sb or self._batch()
        edge = None
        if sb is not None and not general:
            edge, status = mb.sl_edge_cost(sb, self._params())
            if int(status[0]) == _lib.MP_SL_E_CAPACITY:
                edge = None
        if edge is None:
            edge, status = mb.sl_edge_cost_nodes(s_map[None], l_map[None], ego[None], ob[None, :, 0], ob[None, :, 1],
                                                 np.array([len(ob)], np.int32), np.array([ob_dist]), self._params())
        status = int(status[0])
        if status == _lib.MP_SL_E_EMPTY_EDGE:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        if status == _lib.MP_SL_E_CAPACITY:
            raise ValueError("more than 2**20 samples on one lattice edge")
        self._edge_key, self._edge = key, edge
        return edge

    def find_path(self):
        """Fills cost_map / index_map / path_ind_list; False when some layer's
        cheapest node already costs more than 10 * no_path_cost (path_planner.py:66-88)."""
        sb = self._batch()
        out, long_edge = None, False
        if sb is not None and sb.n_l <= 256 and sb.n_s * sb.n_l <= 65536:
            out = mb.sl_plan(sb, self._params())
            if int(out.status[0]) == _lib.MP_SL_E_CAPACITY:  # an edge of more than 128 samples
                out, long_edge = None, True
        if out is None:  # lattices beyond the fused kernel go through the edge tensor
            out = mb.sl_dp(self._edge_tensor(sb, general=long_edge), self._params())
        status = int(out.status[0])
        if status == _lib.MP_SL_E_EMPTY_EDGE:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        self.cost_map = out.cost_map[0].cpu().numpy()
        self.index_map = out.index_map[0].cpu().numpy().astype(np.float64)
        if status != _lib.MP_SL_FOUND:
            print("Path not found!")
            return False
        self.path_ind_list = [int(i) for i in out.path_idx[0].cpu().numpy()]
        return True

    def plan(self):
        """(found, Curve through the ego point and the chosen node of every layer) or (False, [])."""
        if not self.find_path():
            return False, []
        m = self.sl_map
        rows = np.arange(m.n_s)
        path_s = np.concatenate(([m.ego_point[0]], np.asarray(m.s_map)[rows, self.path_ind_list]))
        path_l = np.concatenate(([m.ego_point[1]], np.asarray(m.l_map)[rows, self.path_ind_list]))
        return True, Curve(path_s, self.d_s, path_l, 0.0)

    def get_path(self, p1s, p1l, p2s, p2l, tips):
        """Samples of one edge quintic: zero end slope and curvature, evaluated at
        the absolute arc length as the reference does (path_planner.py:133-155)."""
        poly = QuinticPoly(p1l, 0, 0, p2l, 0, 0, (p2s - p1s))
        ss = get_arange(p1s, p2s, self.d_s)
        if tips == 1:
            return (ss,) + tuple(poly.calc_point(ss, o) for o in range(4))
        return ss, poly.calc_point(ss, 0)

    def cal_cost(self, s_1, l_1, s_2, l_2):
        """Cost of one lattice edge, (s_1, l_1) = (-1, -1) for the ego edge
        (path_planner.py:104-130); read from the GPU edge tensor, which is
        computed once per lattice and reused by later calls."""
        if s_2 != (0 if s_1 == -1 else s_1 + 1):
            raise ValueError("edges connect adjacent layers")
        edge = self._edge_tensor()
        return float(edge[0, s_2, l_2, 0 if s_1 == -1 else l_1])
