"""Batched planning entry points: B independent scenarios per call.

Thin host layer over the C ABI (include/mplattice.h): tensors live in HBM as
torch CUDA tensors (torch is used for device memory and streams only), the
arithmetic is in libmplattice.so.  Every function returns torch tensors on the
same device and enqueues on torch's current stream without synchronising.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import SlBatch, SlParams, check, lib


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("motion_planning_amd needs a GPU: no CUDA/HIP device is visible "
                           "(there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(a, dtype=torch.float64):
    """numpy / list / tensor -> contiguous CUDA tensor of dtype."""
    if isinstance(a, torch.Tensor):
        return a.to(device=_dev(), dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(_dev()).contiguous()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def sl_params(d_s=0.5, w_d=0.5, no_path_cost=1e4, early_exit=True, mid_l=None, n_l=None):
    """Planning/DP_Path/path_planner.py:22-25 constants; mid_l per sl_map.py:20."""
    if mid_l is None:
        mid_l = int((n_l - 1) / 2)
    return SlParams(d_s, w_d, no_path_cost, int(bool(early_exit)), int(mid_l))


class SLBatch:
    """Device-resident SL lattices + obstacles of B scenarios (struct of arrays).

    s_layers [B, n_s]; l_map [B, n_l] (one row shared by all layers) or
    [B, n_s, n_l]; ego_sl [B, 2]; ob_s / ob_l [B, ob_stride]; n_ob [B]; ob_dist [B];
    n_layers [B] (optional): layers scenario b uses out of the n_s slots.
    check=False skips the host-side look at n_ob (no device->host read).
    """

    def __init__(self, s_layers, l_map, ego_sl, ob_s, ob_l, n_ob, ob_dist, n_layers=None, check=True):
        self.s_layers = to_dev(s_layers)
        self.l_map = to_dev(l_map)
        self.ego_sl = to_dev(ego_sl)
        self.B, self.n_s = self.s_layers.shape
        if self.l_map.dim() == 2:
            self.n_l = self.l_map.shape[1]
            self.l_bstride, self.l_rstride = self.n_l, 0
        else:
            assert self.l_map.shape[:2] == (self.B, self.n_s), "l_map must be [B, n_s, n_l]"
            self.n_l = self.l_map.shape[2]
            self.l_bstride, self.l_rstride = self.n_s * self.n_l, self.n_l
        self.ob_s = to_dev(ob_s).reshape(self.B, -1)
        self.ob_l = to_dev(ob_l).reshape(self.B, -1)
        assert self.ob_s.shape == self.ob_l.shape
        self.ob_stride = self.ob_s.shape[1]
        if self.ob_stride == 0:  # keep the pointers valid
            self.ob_s = torch.zeros((self.B, 1), dtype=torch.float64, device=self.ob_s.device)
            self.ob_l = torch.zeros_like(self.ob_s)
        self.n_ob = to_dev(n_ob, torch.int32)
        self.ob_dist = to_dev(ob_dist)
        self.n_layers = to_dev(n_layers, torch.int32) if n_layers is not None else None
        assert self.ego_sl.shape == (self.B, 2) and self.n_ob.shape == (self.B,) and self.ob_dist.shape == (self.B,)
        assert self.n_layers is None or self.n_layers.shape == (self.B,)
        if check:
            assert int(self.n_ob.max()) <= self.ob_stride if self.ob_stride else int(self.n_ob.max()) == 0

    def c_struct(self):
        return SlBatch(self.B, self.n_s, self.n_l, self.ob_stride, _p(self.s_layers), _p(self.l_map),
                       self.l_bstride, self.l_rstride, _p(self.ego_sl), _p(self.ob_s), _p(self.ob_l),
                       _p(self.n_ob), _p(self.ob_dist), _p(self.n_layers))

    def n_edges(self):
        """Lattice edges per scenario: n_l ego edges + (n_s-1)*n_l^2."""
        return self.n_l + (self.n_s - 1) * self.n_l * self.n_l


class SLPlanOut:
    """Pre-allocated outputs of one SL plan call (reusable across calls), and the scratch the
    library orders its workgroups in (heaviest scenario first)."""

    def __init__(self, B, n_s, n_l, want_maps=True, device=None):
        dev = device or _dev()
        self.order_ws = torch.empty(2 * B, dtype=torch.int32, device=dev)
        self.cost_map = torch.empty((B, n_s, n_l), dtype=torch.float64, device=dev) if want_maps else None
        self.index_map = torch.empty((B, n_s, n_l), dtype=torch.int32, device=dev) if want_maps else None
        self.path_idx = torch.empty((B, n_s), dtype=torch.int32, device=dev)
        self.status = torch.empty(B, dtype=torch.int32, device=dev)
        self.layers_done = torch.empty(B, dtype=torch.int32, device=dev)
        self.best_cost = torch.empty(B, dtype=torch.float64, device=dev)


def sl_plan(batch, params=None, out=None, want_maps=True, ordered=True):
    """Fused SL path DP (replaces PathPlanner.find_path, path_planner.py:66-101).
    ordered=False launches the scenarios in the caller's order (mp_sl_plan) instead of
    heaviest first (mp_sl_plan_ordered); the results are the same."""
    params = params or sl_params(n_l=batch.n_l)
    out = out or SLPlanOut(batch.B, batch.n_s, batch.n_l, want_maps)
    cs = batch.c_struct()
    if ordered and out.order_ws.numel() >= 2 * batch.B:
        check(lib().mp_sl_plan_ordered(C.byref(cs), C.byref(params), _p(out.cost_map), _p(out.index_map),
                                       _p(out.path_idx), _p(out.status), _p(out.layers_done), _p(out.best_cost),
                                       _p(out.order_ws), _stream()), "mp_sl_plan_ordered")
        return out
    check(lib().mp_sl_plan(C.byref(cs), C.byref(params), _p(out.cost_map), _p(out.index_map), _p(out.path_idx),
                           _p(out.status), _p(out.layers_done), _p(out.best_cost), _stream()), "mp_sl_plan")
    return out


def sl_plan_counted(batch, params=None, out=None):
    """mp_sl_plan through the counting instance of the fused kernel (measurement aid):
    returns (out, edges the kernel priced over the whole batch, edges priced one after the
    other: the busiest lane of every wave, layer by layer).  Reads the counters back, so it
    synchronises."""
    params = params or sl_params(n_l=batch.n_l)
    out = out or SLPlanOut(batch.B, batch.n_s, batch.n_l, False)
    n_eval = torch.zeros(4, dtype=torch.int64, device=batch.s_layers.device)
    cs = batch.c_struct()
    check(lib().mp_sl_plan_counted(C.byref(cs), C.byref(params), _p(out.path_idx), _p(out.status),
                                   _p(out.layers_done), _p(out.best_cost), _p(n_eval), _stream()),
          "mp_sl_plan_counted")
    n_eval = n_eval.cpu()
    out.walk_turns, out.walk_tests = int(n_eval[2]), int(n_eval[3])   # busiest lane of every wave, layer by layer
    return out, int(n_eval[0]), int(n_eval[1])


def sl_edge_cost(batch, params=None):
    """Every lattice edge cost (replaces PathPlanner.cal_cost, path_planner.py:104-130).

    Returns (edge_cost [B, n_s, n_l(to), n_l(from)], status [B])."""
    params = params or sl_params(n_l=batch.n_l)
    dev = batch.s_layers.device
    edge = torch.empty((batch.B, batch.n_s, batch.n_l, batch.n_l), dtype=torch.float64, device=dev)
    status = torch.empty(batch.B, dtype=torch.int32, device=dev)
    cs = batch.c_struct()
    check(lib().mp_sl_edge_cost(C.byref(cs), C.byref(params), _p(edge), _p(status), _stream()), "mp_sl_edge_cost")
    return edge, status


def sl_edge_cost_nodes(s_map, l_map, ego_sl, ob_s, ob_l, n_ob, ob_dist, params=None, n_layers=None):
    """Edge costs of lattices whose nodes carry their own arc length (the reference reads
    s_map[s][l] per node, path_planner.py:104-123), any number of samples per edge.

    s_map / l_map [B, n_s, n_l]; ego_sl [B, 2]; ob_s / ob_l [B, ob_stride]; n_ob, ob_dist [B].
    Returns (edge_cost [B, n_s, n_l(to), n_l(from)], status [B]) as sl_edge_cost does."""
    s_map, l_map, ego_sl = to_dev(s_map), to_dev(l_map), to_dev(ego_sl)
    B, n_s, n_l = s_map.shape
    assert l_map.shape == (B, n_s, n_l) and ego_sl.shape == (B, 2)
    ob_s, ob_l = to_dev(ob_s).reshape(B, -1), to_dev(ob_l).reshape(B, -1)
    assert ob_s.shape == ob_l.shape
    ob_stride = ob_s.shape[1]
    n_ob, ob_dist = to_dev(n_ob, torch.int32), to_dev(ob_dist)
    assert n_ob.shape == (B,) and ob_dist.shape == (B,)
    assert int(n_ob.max()) <= ob_stride
    n_layers = to_dev(n_layers, torch.int32) if n_layers is not None else None
    params = params or sl_params(n_l=n_l)
    edge = torch.empty((B, n_s, n_l, n_l), dtype=torch.float64, device=s_map.device)
    status = torch.empty(B, dtype=torch.int32, device=s_map.device)
    check(lib().mp_sl_edge_cost_nodes(B, n_s, n_l, _p(s_map), _p(l_map), _p(ego_sl),
                                      _p(ob_s if ob_stride else None), _p(ob_l if ob_stride else None), ob_stride,
                                      _p(n_ob), _p(ob_dist), _p(n_layers), C.byref(params), _p(edge), _p(status),
                                      _stream()), "mp_sl_edge_cost_nodes")
    return edge, status


def sl_dp(edge_cost, params, out=None, n_layers=None):
    """Layer-wise min-relaxation over materialised edge costs (path_planner.py:66-101)."""
    B, n_s, n_l, _ = edge_cost.shape
    out = out or SLPlanOut(B, n_s, n_l, True, edge_cost.device)
    n_layers = to_dev(n_layers, torch.int32) if n_layers is not None else None
    check(lib().mp_sl_dp(B, n_s, n_l, _p(n_layers), _p(edge_cost), C.byref(params), _p(out.cost_map),
                         _p(out.index_map),
                         _p(out.path_idx), _p(out.status), _p(out.layers_done), _p(out.best_cost), _stream()),
          "mp_sl_dp")
    return out


def st_dp(n_s, s_down, s_up, n_reg, cur_vel, speed_lim, ds=1.0, dt=0.5, want_tables=True, n_s_max=None, out=None,
          ordered=False):
    """ST speed DP (replaces SpeedPlanner.find_speed, speed_planner.py:57-162).

    n_s [B]; s_down / s_up [B, n_t, r_max]; n_reg [B, n_t]; cur_vel [B].
    Returns dict(cost, pre, path_s_ind [B, n_t], path_len [B]); `out` (a dict returned by
    an earlier call of the same shape) is written in place instead of allocating.
    ordered=True launches the widest grids first (mp_st_dp_ordered; for batches whose grids
    differ in width; same results)."""
    n_s = to_dev(n_s, torch.int32)
    n_reg = to_dev(n_reg, torch.int32)
    B, n_t = n_reg.shape
    s_down = to_dev(s_down).reshape(B, n_t, -1)
    s_up = to_dev(s_up).reshape(B, n_t, -1)
    r_max = s_down.shape[2]
    cur_vel = to_dev(cur_vel)
    N = int(n_s_max if n_s_max is not None else n_s.max())
    dev = n_s.device
    if out is not None:
        cost, pre, idx, plen = out["cost"], out["pre"], out["path_s_ind"], out["path_len"]
        assert idx.shape == (B, n_t) and plen.shape == (B,) and (cost is None or cost.shape == (B, n_t, N))
    else:
        cost = torch.empty((B, n_t, N), dtype=torch.float64, device=dev) if want_tables else None
        pre = torch.empty((B, n_t, N), dtype=torch.int32, device=dev) if want_tables else None
        idx = torch.empty((B, n_t), dtype=torch.int32, device=dev)
        plen = torch.empty(B, dtype=torch.int32, device=dev)
    if ordered:
        ws = out.get("order_ws") if out is not None else None
        if ws is None or ws.numel() < 2 * B:
            ws = torch.empty(2 * B, dtype=torch.int32, device=dev)
        check(lib().mp_st_dp_ordered(B, n_t, N, _p(n_s), _p(s_down), _p(s_up), _p(n_reg), r_max, _p(cur_vel),
                                     float(speed_lim), float(ds), float(dt), _p(cost), _p(pre), _p(idx), _p(plen),
                                     _p(ws), _stream()), "mp_st_dp_ordered")
        return dict(cost=cost, pre=pre, path_s_ind=idx, path_len=plen, order_ws=ws)
    check(lib().mp_st_dp(B, n_t, N, _p(n_s), _p(s_down), _p(s_up), _p(n_reg), r_max, _p(cur_vel),
                         float(speed_lim), float(ds), float(dt), _p(cost), _p(pre), _p(idx), _p(plen),
                         _stream()), "mp_st_dp")
    return dict(cost=cost, pre=pre, path_s_ind=idx, path_len=plen)


def st_regions(n_t, path_s, path_l, n_path, st_s, st_l, st_r, n_st, dy_s, dy_l, dy_r, n_dy, r_max):
    """ST blocked intervals (replaces STMap.add_obstacle, st_map.py:24-60).

    path_s / path_l [B, p_max]; statics [B, n_st_max]; dynamics dy_s / dy_l
    [B, n_dy_max, n_t], dy_r [B, n_dy_max].  Returns (s_down, s_up [B, n_t, r_max],
    n_reg [B, n_t], status [B])."""
    path_s, path_l = to_dev(path_s), to_dev(path_l)
    B, p_max = path_s.shape
    n_path = to_dev(n_path, torch.int32)
    def _2d(x, tail=()):
        x = to_dev(x)
        return x.reshape((B, -1) + tail) if x.numel() else x.new_zeros((B, 0) + tail)

    st_s, st_l, st_r = (_2d(x) for x in (st_s, st_l, st_r))
    n_st = to_dev(n_st, torch.int32)
    n_st_max = st_s.shape[1]
    dy_s, dy_l = (_2d(x, (n_t,)) for x in (dy_s, dy_l))
    dy_r = _2d(dy_r)
    n_dy = to_dev(n_dy, torch.int32)
    n_dy_max = dy_r.shape[1]
    dev = path_s.device
    s_down = torch.zeros((B, n_t, r_max), dtype=torch.float64, device=dev)
    s_up = torch.zeros((B, n_t, r_max), dtype=torch.float64, device=dev)
    n_reg = torch.empty((B, n_t), dtype=torch.int32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    check(lib().mp_st_regions(B, n_t, _p(path_s), _p(path_l), _p(n_path), p_max,
                              _p(st_s) if n_st_max else None, _p(st_l) if n_st_max else None,
                              _p(st_r) if n_st_max else None, _p(n_st) if n_st_max else None, n_st_max,
                              _p(dy_s) if n_dy_max else None, _p(dy_l) if n_dy_max else None,
                              _p(dy_r) if n_dy_max else None, _p(n_dy) if n_dy_max else None, n_dy_max,
                              r_max, _p(s_down), _p(s_up), _p(n_reg), _p(status), _stream()), "mp_st_regions")
    return s_down, s_up, n_reg, status


class RefLineTables:
    """Per-scenario reference-line sample tables in HBM (output of refline_tables)."""

    def __init__(self, B, m_max, device):
        f = dict(dtype=torch.float64, device=device)
        self.B, self.m_max = B, m_max
        self.tab_x = torch.zeros((B, m_max), **f)
        self.tab_y = torch.zeros((B, m_max), **f)
        self.tab_s = torch.zeros((B, m_max), **f)
        self.tab_theta = torch.zeros((B, m_max), **f)
        self.tab_kappa = torch.zeros((B, m_max), **f)
        self.M = torch.zeros(B, dtype=torch.int32, device=device)
        self.origin = torch.zeros((B, 4), **f)   # rx_ori, ry_ori, get_s(last knot), rx_max
        self.status = torch.zeros(B, dtype=torch.int32, device=device)
        # Frenet -> Cartesian table (abscissa, arc length, ordinate, heading)
        self.inv_x = torch.zeros((B, m_max), **f)
        self.inv_s = torch.zeros((B, m_max), **f)
        self.inv_y = torch.zeros((B, m_max), **f)
        self.inv_theta = torch.zeros((B, m_max), **f)
        self.n_inv = torch.zeros(B, dtype=torch.int32, device=device)


def refline_tables(knots_x, knots_y, n_knots, vec, ego_xy, d_x=0.5, m_max=None):
    """Reference-line tables (replaces CartesianFrenetConverter.__init__ and the
    per-sample parts of cartesian_to_frenet / frenet_to_cartesian,
    cartesian_frenet_conversion.py:11-86).

    knots_x / knots_y [B, n_knots_max]; n_knots [B]; vec [B]; ego_xy [B, 2]."""
    knots_x, knots_y = to_dev(knots_x), to_dev(knots_y)
    B, n_knots_max = knots_x.shape
    n_knots = to_dev(n_knots, torch.int32)
    vec, ego_xy = to_dev(vec), to_dev(ego_xy)
    if m_max is None:
        span = float((knots_x.max(dim=1).values - knots_x.min(dim=1).values).max())
        m_max = int(span / d_x) + 8
    t = RefLineTables(B, m_max, knots_x.device)
    check(lib().mp_refline_tables(B, _p(knots_x), _p(knots_y), _p(n_knots), n_knots_max, _p(vec), _p(ego_xy),
                                  float(d_x), m_max, _p(t.tab_x), _p(t.tab_y), _p(t.tab_s), _p(t.tab_theta),
                                  _p(t.tab_kappa), _p(t.M), _p(t.origin), _p(t.status), _p(t.inv_x), _p(t.inv_s),
                                  _p(t.inv_y), _p(t.inv_theta), _p(t.n_inv), _stream()),
          "mp_refline_tables")
    return t


def c2f(tables, px, py, pt_scn, vx=None, vy=None):
    """Batched cartesian_to_frenet (cartesian_frenet_conversion.py:32-53).

    px / py [n_pts]; pt_scn [n_pts] scenario of each point; with vx / vy also s_dt.
    Returns (s, l) or (s, l, s_dt)."""
    px, py = to_dev(px).reshape(-1), to_dev(py).reshape(-1)
    pt_scn = to_dev(pt_scn, torch.int32).reshape(-1)
    n = px.numel()
    s = torch.empty(n, dtype=torch.float64, device=px.device)
    l = torch.empty_like(s)
    order1 = vx is not None
    if order1:
        vx, vy = to_dev(vx).reshape(-1), to_dev(vy).reshape(-1)
    s_dt = torch.empty_like(s) if order1 else None
    check(lib().mp_c2f_batch(n, _p(px), _p(py), _p(vx) if order1 else None, _p(vy) if order1 else None,
                             _p(pt_scn), tables.m_max, _p(tables.tab_x), _p(tables.tab_y), _p(tables.tab_s),
                             _p(tables.tab_theta), _p(tables.tab_kappa), _p(tables.M), _p(s), _p(l), _p(s_dt),
                             _stream()), "mp_c2f_batch")
    return (s, l, s_dt) if order1 else (s, l)


def f2c(tables, s, l, pt_scn):
    """Batched frenet_to_cartesian (cartesian_frenet_conversion.py:55-60, 72-86)."""
    s, l = to_dev(s).reshape(-1), to_dev(l).reshape(-1)
    pt_scn = to_dev(pt_scn, torch.int32).reshape(-1)
    n = s.numel()
    x = torch.empty(n, dtype=torch.float64, device=s.device)
    y = torch.empty_like(x)
    check(lib().mp_f2c_batch(n, _p(s), _p(l), _p(pt_scn), tables.m_max, _p(tables.inv_x), _p(tables.inv_s),
                             _p(tables.inv_y), _p(tables.inv_theta), _p(tables.n_inv), _p(x), _p(y), _stream()),
          "mp_f2c_batch")
    return x, y


CURVE_ARC, CURVE_KNOTS, CURVE_QUERY = 0, 1, 2


def curve_eval(knots_t, knots_x, n_knots, vec, dt, mode, q_max, query=None, n_query=None, want_dx=False):
    """Batched spline build + sampling (Curve.__init__ + calc_point_arr, curve.py:13-73).

    mode CURVE_ARC: samples every dt over [t0, t0 + sum(T)] (SLMap.path_sampling);
    CURVE_KNOTS: over [t0, last knot] (SpeedPlanner.plan); CURVE_QUERY: at query
    [B, q_max] (first n_query[b]).  Returns dict(t, x[, dx], n, status)."""
    knots_t, knots_x = to_dev(knots_t), to_dev(knots_x)
    B, nk = knots_t.shape
    n_knots, vec = to_dev(n_knots, torch.int32), to_dev(vec)
    dev = knots_t.device
    if mode == CURVE_QUERY:
        query = to_dev(query).reshape(B, -1)
        assert query.shape[1] == q_max
        n_query = to_dev(n_query, torch.int32)
    out_t = torch.zeros((B, q_max), dtype=torch.float64, device=dev)
    out_x = torch.zeros_like(out_t)
    out_dx = torch.zeros_like(out_t) if want_dx else None
    n_out = torch.empty(B, dtype=torch.int32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    check(lib().mp_curve_eval(B, _p(knots_t), _p(knots_x), _p(n_knots), nk, _p(vec), float(dt), int(mode),
                              _p(query), _p(n_query), q_max, _p(out_t), _p(out_x), _p(out_dx), _p(n_out),
                              _p(status), _stream()), "mp_curve_eval")
    return dict(t=out_t, x=out_x, dx=out_dx, n=n_out, status=status)


N_FRAMES = 11  # predicted frames of a moving obstacle: t = 0, 0.5, ..., 5.0 (Model/obstacle.py:46-53)


def scene_ingest(rot, trans, ego_vel_w, way_xy, n_way, ob_pos, ob_vel, ob_rad, n_ob, ignore_dist=20.0,
                 n_frames=N_FRAMES, dt=0.5):
    """Batched RobotMap.add_robot / add_ref_line / add_obstacle (robot_map.py:25-68):
    world-frame scenes into the ego frame, obstacles filtered, classified and rolled out.

    rot [B, 2, 2] (or [B, 4]) ego rotation, trans [B, 2], ego_vel_w [B, 2]; way_xy [B, W, 2],
    n_way [B]; ob_pos / ob_vel [B, O, 2], ob_rad [B, O], n_ob [B].  Returns a dict with
    knots_x, knots_y [B, W], ego_vel [B, 2], st_pos [B, O, 2], st_rad, n_st, dy_pos
    [B, O, n_frames, 2], dy_vel [B, O, 2], dy_rad, n_dy, ob_class [B, O]."""
    rot, trans, ego_vel_w = to_dev(rot).reshape(-1, 4), to_dev(trans).reshape(-1, 2), to_dev(ego_vel_w).reshape(-1, 2)
    B = rot.shape[0]
    way_xy = to_dev(way_xy).reshape(B, -1, 2)
    W = way_xy.shape[1]
    n_way, n_ob = to_dev(n_way, torch.int32), to_dev(n_ob, torch.int32)
    ob_pos, ob_vel = to_dev(ob_pos).reshape(B, -1, 2), to_dev(ob_vel).reshape(B, -1, 2)
    ob_rad = to_dev(ob_rad).reshape(B, -1)
    O = ob_pos.shape[1]
    assert ob_vel.shape[1] == O and ob_rad.shape[1] == O
    dev = rot.device
    f64 = dict(dtype=torch.float64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    out = dict(knots_x=torch.empty((B, W), **f64), knots_y=torch.empty((B, W), **f64),
               ego_vel=torch.empty((B, 2), **f64), st_pos=torch.zeros((B, O, 2), **f64),
               st_rad=torch.zeros((B, O), **f64), n_st=torch.empty(B, **i32),
               dy_pos=torch.zeros((B, O, n_frames, 2), **f64), dy_vel=torch.zeros((B, O, 2), **f64),
               dy_rad=torch.zeros((B, O), **f64), n_dy=torch.empty(B, **i32),
               ob_class=torch.full((B, O), -1, **i32))
    check(lib().mp_scene_ingest(B, _p(rot), _p(trans), _p(ego_vel_w), _p(way_xy), _p(n_way), W,
                                _p(ob_pos) if O else None, _p(ob_vel) if O else None, _p(ob_rad) if O else None,
                                _p(n_ob), O, float(ignore_dist), int(n_frames), float(dt), _p(out["knots_x"]),
                                _p(out["knots_y"]), _p(out["ego_vel"]), _p(out["st_pos"]) if O else None,
                                _p(out["st_rad"]) if O else None, _p(out["n_st"]),
                                _p(out["dy_pos"]) if O else None, _p(out["dy_vel"]) if O else None,
                                _p(out["dy_rad"]) if O else None, _p(out["n_dy"]),
                                _p(out["ob_class"]) if O else None, _stream()), "mp_scene_ingest")
    return out


def robot_to_world(rot, trans, x, y, pt_scn):
    """Batched RobotMap.path_convert (robot_map.py:84-91): R p + t per point."""
    rot, trans = to_dev(rot).reshape(-1, 4), to_dev(trans).reshape(-1, 2)
    x, y = to_dev(x).reshape(-1), to_dev(y).reshape(-1)
    pt_scn = to_dev(pt_scn, torch.int32).reshape(-1)
    n = x.numel()
    wx, wy = torch.empty_like(x), torch.empty_like(y)
    check(lib().mp_robot_to_world(n, _p(x), _p(y), _p(pt_scn), _p(rot), _p(trans), _p(wx), _p(wy), _stream()),
          "mp_robot_to_world")
    return wx, wy
