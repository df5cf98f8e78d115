#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container, where the reference checkout is mounted at
/root/reference; the GPU box never sees it.  The reference is imported with a
stub `carla` module (Utils/tool.py imports carla at module top) and with the
reference's own directories on sys.path (it uses Python-2 implicit relative
imports).  Only inputs and outputs are written: no reference source travels.

    python tools/gen_golden.py            # rewrites tests/golden/*.npz
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REF = os.environ.get("MP_REFERENCE", "/root/reference")
sys.path[:0] = [REF, os.path.join(REF, "Model"), os.path.join(REF, "Planning", "Interface")]
sys.modules["carla"] = types.ModuleType("carla")
import matplotlib  # noqa: E402

matplotlib.use("Agg")

from robot_map import RobotMap  # noqa: E402
from Model.curve import Curve, QuinticPoly  # noqa: E402
from Model.cartesian_frenet_conversion import CartesianFrenetConverter  # noqa: E402
from Planning.DP_Path.sl_map import SLMap  # noqa: E402
from Planning.DP_Path.path_planner import PathPlanner  # noqa: E402
from Planning.DP_Path.track_path_planner import TrackPathPlanner  # noqa: E402
from Planning.DP_Speed.st_map import STMap, ObState  # noqa: E402
from Planning.DP_Speed.speed_planner import SpeedPlanner  # noqa: E402
from Utils.tool import get_arange  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


class DuckSL:
    """Stands in for SLMap: PathPlanner reads only these attributes."""

    def __init__(self, s_layers, l_row, ego, ob, ob_dist):
        self.n_s, self.n_l = len(s_layers), len(l_row)
        self.mid_l = int((self.n_l - 1) / 2)
        self.s_map = np.array([s * np.ones(self.n_l) for s in s_layers])
        self.l_map = np.array([np.array(l_row, dtype=float) for _ in s_layers])
        self.ego_point = np.array(ego, dtype=float)
        self.ob_list = [np.array(o, dtype=float) for o in ob]
        self.ob_dist = ob_dist


def run_sl(duck, want_edges=False):
    pp = PathPlanner(duck)
    found = quiet(pp.find_path)
    out = dict(found=np.array(found), cost_map=pp.cost_map.copy(), index_map=pp.index_map.astype(np.int32),
               path_idx=np.array(pp.path_ind_list, dtype=np.int32) if found else np.zeros(0, np.int32))
    if want_edges:
        e = np.full((duck.n_s, duck.n_l, duck.n_l), np.nan)
        for l in range(duck.n_l):
            e[0, l, 0] = pp.cal_cost(-1, -1, 0, l)
        for s in range(1, duck.n_s):
            for l in range(duck.n_l):
                for j in range(duck.n_l):
                    e[s, l, j] = pp.cal_cost(s - 1, j, s, l)
        out["edge_cost"] = e
    return out


def duck_inputs(d):
    return dict(s_map=d.s_map, l_map=d.l_map, ego=d.ego_point, ob=np.array(d.ob_list).reshape(-1, 2),
                ob_dist=np.array(d.ob_dist), mid_l=np.array(d.mid_l))


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, sum(np.asarray(v).nbytes for v in arrs.values()), "bytes raw")


def prefixed(prefix, d):
    return {prefix + k: v for k, v in d.items()}


# ---------------------------------------------------------------- primitives
def gen_primitives():
    rng = np.random.default_rng(7)
    out = {}
    # get_arange cases (Utils/tool.py:27-33)
    cases = [(0.0, 8.0, 0.5), (8.0, 16.0, 0.5), (16.0, 28.0, 0.5), (1.5, 3.0, 0.5), (0.0, 0.2, 0.5),
             (0.0, 0.5, 0.5), (0.3, 7.9, 0.5), (0.0, 5.0, 0.1), (0.0, 4.5, 0.1), (2.0, 26.73, 0.5),
             (0.1, 0.7, 0.1), (0.0, 1.0, 0.1), (0.0, 0.3, 0.1)]
    cases += [(float(a), float(a + b), 0.5) for a, b in zip(rng.uniform(0, 50, 20), rng.uniform(0, 30, 20))]
    out["ar_in"] = np.array(cases)
    vals = [get_arange(*c) for c in cases]
    out["ar_len"] = np.array([len(v) for v in vals])
    out["ar_val"] = np.concatenate(vals)
    # QuinticPoly (Model/curve.py:129-173)
    qin = np.concatenate([rng.uniform(-2, 2, (12, 6)), rng.uniform(0.5, 9, (12, 1))], axis=1)
    qin[:4, [1, 2, 4, 5]] = 0.0
    tq = np.concatenate([np.linspace(0, 9, 10), [16.0, 31.5, 64.0]])
    coef, ev = [], []
    for r in qin:
        q = QuinticPoly(*r)
        coef.append([q.a0, q.a1, q.a2, q.a3, q.a4, q.a5])
        ev.append([q.calc_point(tq, o) for o in range(4)])
    out.update(q_in=qin, q_t=tq, q_coef=np.array(coef), q_eval=np.array(ev))
    # Curve (Model/curve.py:13-126), including the lagged segment switch
    curves = [
        (np.array([0.0, 2, 4, 6]), 0.5, np.array([0.0, 1, 0, 2]), 0.0),
        (np.array([-2 + 4.5 * i for i in range(7)]), 0.5, np.zeros(7), 0.0),
        (np.array([-2 + 4.5 * i for i in range(9)]), 0.5, 0.004 * np.array([-2 + 4.5 * i for i in range(9)]) ** 2, 0.0),
        (np.array([0.0, 0.5, 1.0, 1.5, 2.0, 2.5]), 0.1, np.array([0.0, 1, 3, 4, 6, 9]), 5.0),
        (np.array([0.0, 8.0, 16.0, 27.3]), 0.5, np.array([-0.1, 0.725, 0.725, -1.45]), 0.0),
        (np.array([1.0, 3.5]), 0.5, np.array([0.3, -0.4]), 0.2),
    ]
    for ci, (t, dt, x, vec) in enumerate(curves):
        cv = Curve(t, dt, x, vec)
        tt = get_arange(0, np.sum(cv.T), dt) + cv.t_bios
        p = "cv%d_" % ci
        out[p + "t"], out[p + "x"], out[p + "dtvec"] = t, x, np.array([dt, vec])
        out[p + "tt"] = tt
        out[p + "arr"] = np.array([cv.calc_point_arr(tt, o) for o in range(4)])
        out[p + "scalar"] = np.array([[cv.calc_point(u, o) for u in tt] for o in range(3)])
        out[p + "arc"] = np.array([cv.calc_arc_len(cv.t_bios, u) for u in tt])
        out[p + "arc_mid"] = np.array([cv.calc_arc_len(tt[len(tt) // 3], u) for u in tt])
        pts = rng.uniform([t[0] - 1, -3], [t[-1] + 1, 3], (16, 2))
        pr = [cv.projection(a, b) for a, b in pts]
        out[p + "proj_in"] = pts
        out[p + "proj_out"] = np.array([[d, q[0], q[1]] for d, q in pr])
    out["n_curves"] = np.array(len(curves))
    save("primitives.npz", **out)


# ---------------------------------------------------------------- C2F
def ref_lines():
    xs7 = np.array([-2 + 4.5 * i for i in range(7)])
    xs16 = np.array([-2 + 4.5 * i for i in range(16)])
    xs20 = np.array([-3.1 + 4.5 * i for i in range(20)])
    return [(xs7, np.zeros(7)), (xs16, 0.004 * xs16 ** 2), (xs20, -0.006 * xs20 ** 2 + 0.05 * xs20),
            (xs7, 0.02 * xs7 ** 2)]


def gen_c2f():
    rng = np.random.default_rng(11)
    out = {}
    lines = ref_lines()
    for li, (rx, ry) in enumerate(lines):
        conv = quiet(CartesianFrenetConverter, 0.0, 0.0, rx, ry, 0.0)
        p = "l%d_" % li
        pts = rng.uniform([rx[0] - 3, -6], [rx[-1] + 3, 6], (64, 2))
        pts[:4, 1] = np.interp(pts[:4, 0], rx, ry)   # near / on the line
        if li == 0:
            pts[:6, 1] = 0.0                          # exactly on a straight line
        pts[4:8, 0] = rng.uniform(rx[0] - 3, -0.5, 4)  # behind the origin
        vel = rng.uniform(-6, 8, (64, 2))
        o0 = np.array([quiet(conv.cartesian_to_frenet, x, y) for x, y in pts])
        o1 = np.array([quiet(conv.cartesian_to_frenet, x, y, vx, vy, 1) for (x, y), (vx, vy) in zip(pts, vel)])
        tt = get_arange(0, np.sum(conv.ref_curve.T), 0.5) + conv.ref_curve.t_bios
        out[p + "rx"], out[p + "ry"] = rx, ry
        out[p + "ori"] = np.array([conv.rx_ori, conv.ry_ori, conv.rx_max])
        out[p + "pts"], out[p + "vel"], out[p + "o0"], out[p + "o1"] = pts, vel, o0, o1
        out[p + "tab_x"] = tt
        out[p + "tab_y"] = conv.ref_curve.calc_point_arr(tt, 0)
        out[p + "tab_S"] = np.array([quiet(conv.get_s, u) for u in tt])
        out[p + "tab_th"] = np.array([np.arctan(conv.ref_curve.calc_point(u, 1)) for u in tt])
        out[p + "tab_kr"] = np.array([conv.get_curvature(u) for u in tt])
        sq = np.concatenate([[0.0, 0.3], rng.uniform(0, out[p + "tab_S"][-1] + 2, 10)])
        lq = rng.uniform(-1.5, 1.5, len(sq))
        out[p + "f2c_in"] = np.stack([sq, lq], 1)
        out[p + "f2c_out"] = np.array([quiet(conv.frenet_to_cartesian, a, b) for a, b in zip(sq, lq)])
    out["n_lines"] = np.array(len(lines))
    save("c2f.npz", **out)


# ---------------------------------------------------------------- SL
def kat1_robot_map():
    rm = RobotMap(np.eye(2), np.zeros(2))
    rm.add_robot(np.zeros(2), np.array([5.0, 0.0]), 0.0)
    knots = [np.array([-2 + 4.5 * i, 0.0]) for i in range(7)]
    rm.add_ref_line(knots, 3.5, 5, len(knots), 0)
    for p in [(8, 0.2), (14, -1.0), (19, 1.2)]:
        quiet(rm.add_obstacle, np.array(p, dtype=float), np.zeros(2), 1.0)
    return rm


def gen_sl():
    rng = np.random.default_rng(23)
    out = {}
    # cfg1 / KAT-1 through the real SLMap
    rm = kat1_robot_map()
    sl = quiet(SLMap, rm)
    r = run_sl(sl, want_edges=True)
    out.update(prefixed("kat1_", duck_inputs(sl)))
    out.update(prefixed("kat1_", r))
    out["kat1_robot_vel"] = np.array(sl.robot_vel)
    found, curve = quiet(PathPlanner(sl).plan)
    pb = sl.path_sampling(curve)
    out["kat1_path_sl"] = np.array(pb)
    out["kat1_curve_T"] = curve.T
    # duck-typed lattices
    cases = {}
    lin = lambda n: np.linspace(-1.45, 1.45, n)  # noqa: E731

    def rand_ob(n, s_hi, lw=2.5):
        return np.stack([rng.uniform(2, s_hi, n), rng.uniform(-lw, lw, n)], 1)

    cases["d8x16"] = DuckSL([8.0 * (k + 1) for k in range(8)], lin(16), (0.0, 0.13), rand_ob(10, 64), 0.7)
    cases["d12x5"] = DuckSL([8.0 * (k + 1) for k in range(11)] + [93.7], lin(5), (0.0, -0.2), rand_ob(3, 90), 1.0)
    cases["d16x32nf"] = DuckSL([1.5 * (k + 1) for k in range(16)], lin(32), (0.0, 0.05), rand_ob(10, 24, 1.4), 0.8)
    cases["d10x12"] = DuckSL([1.5 * (k + 1) for k in range(10)], lin(12), (0.0, -0.21), rand_ob(4, 15), 0.35)
    wall = [(s0, l0) for s0 in (6.0, 12.0, 18.0, 21.0) for l0 in np.arange(-2.0, 2.01, 0.5)]
    cases["wall_nf"] = DuckSL([1.5 * (k + 1) for k in range(20)], lin(12), (0.0, 0.1), wall, 0.6)
    dense = np.stack([rng.uniform(0.5, 9, 12), rng.uniform(-1.6, 1.6, 12)], 1)
    cases["d12x16dense"] = DuckSL([1.5 * (k + 1) for k in range(12)], lin(16), (0.0, 0.4), dense, 0.3)
    dense2 = np.stack([rng.uniform(1, 30, 9), rng.uniform(-1.6, 1.6, 9)], 1)
    cases["d6x9dense"] = DuckSL([8.0 * (k + 1) for k in range(6)], lin(9), (0.0, -0.6), dense2, 0.7)
    cases["tie5"] = DuckSL([8.0, 16.0, 24.0, 35.0], lin(5), (0.0, 0.0), [], 0.0)
    cases["tie6"] = DuckSL([8.0, 16.0, 24.0], lin(6), (0.0, 0.0), [], 0.0)
    cases["ragged"] = DuckSL([8.0, 16.0, 24.3, 30.9], lin(7), (0.4, 0.31), rand_ob(5, 30), 0.6)
    cases["egoback"] = DuckSL([3.0, 6.2, 9.0], lin(9), (0.0, 1.0), [(4.0, 0.9), (7.0, -0.3)], 0.5)
    for name, d in cases.items():
        r = run_sl(d, want_edges=(d.n_s * d.n_l * d.n_l <= 4096))
        out.update(prefixed(name + "_", duck_inputs(d)))
        out.update(prefixed(name + "_", r))
        print(name, "found", r["found"], r["path_idx"])
    out["cases"] = np.array(["kat1"] + list(cases.keys()))
    # isolated edges incl. p1s >> 0 (absolute-s evaluation) with parts
    E = 48
    ed_in = np.zeros((E, 4))
    ed_in[:, 0] = rng.choice([0.0, 1.5, 8.0, 16.0, 40.0, 94.5], E)
    ed_in[:, 2] = ed_in[:, 0] + rng.choice([1.5, 8.0, 5.3, 11.9], E)
    ed_in[:, 1] = rng.uniform(-1.45, 1.45, E)
    ed_in[:, 3] = rng.uniform(-1.45, 1.45, E)
    ed_in[::6, 3] = ed_in[::6, 1]
    ob = rand_ob(6, 100)
    ed_cost = []
    for p1s, p1l, p2s, p2l in ed_in:
        d = DuckSL([p1s + 1e-9 if p1s == 0 else p1s, p2s], [p1l, p2l, 0.0], (0.0, p1l), ob, 0.9)
        d.mid_l = 2
        d.l_map[0, 0] = p1l
        d.s_map[0, :] = p1s
        pp = PathPlanner(d)
        ed_cost.append(pp.cal_cost(0, 0, 1, 1))
    out.update(edge_in=ed_in, edge_ob=ob, edge_ob_dist=np.array(0.9), edge_cost=np.array(ed_cost))
    save("sl.npz", **out)


# ---------------------------------------------------------------- ST
def kat2_regions(n_t, scale=1.0):
    base = [(12, 2), (20, 0), (6, 4), (25, -1), (15, 1)]
    down = np.zeros((n_t, 5))
    up = np.zeros((n_t, 5))
    for i in range(n_t):
        for k, (s0, v) in enumerate(base):
            down[i, k] = s0 * scale + v * 0.5 * i
            up[i, k] = down[i, k] + 3
    return down, up


def run_st(s_end, down, up, n_reg, cur_vel, lim):
    st = STMap(None, s_end, 5)
    for i in range(st.n_t):
        o = ObState(i * st.dt)
        o.s_down = list(down[i, :n_reg[i]])
        o.s_up = list(up[i, :n_reg[i]])
        st.ob_mat.append(o)
    sp = SpeedPlanner(st, cur_vel, lim)
    ss, vv = quiet(sp.plan)
    cost = np.array([[st.map[i][j].cost for j in range(st.n_s)] for i in range(st.n_t)])
    pre = np.array([[st.map[i][j].pre_ind for j in range(st.n_s)] for i in range(st.n_t)], dtype=np.int32)
    return dict(n_s=np.array(st.n_s), n_t=np.array(st.n_t), cost=cost, pre=pre,
                path_s_ind=np.array(sp.path_s_ind, dtype=np.int32), ss=ss, vv=vv)


def gen_st():
    rng = np.random.default_rng(31)
    out = {}
    lim = 25.0 * 5.0 / 18.0
    names = []
    for s_end, scale in [(28.0, 1.0), (60.0, 2.0), (100.0, 3.5)]:
        down, up = kat2_regions(11, scale)
        n_reg = np.full(11, 5, dtype=np.int32)
        name = "kat2_%d" % int(s_end)
        r = run_st(s_end, down, up, n_reg, 5.0, lim)
        out.update(prefixed(name + "_", dict(s_end=np.array(s_end), down=down, up=up, n_reg=n_reg,
                                             cur_vel=np.array(5.0), lim=np.array(lim), **r)))
        names.append(name)
    # random ragged regions, odd start speeds, no regions, fast start
    for ci in range(6):
        s_end = float(rng.uniform(12, 70))
        n_reg = rng.integers(0, 5, 11).astype(np.int32)
        down = rng.uniform(2, s_end, (11, 4))
        up = down + rng.uniform(0, 6, (11, 4))
        cur = float(rng.uniform(0, 9))
        if ci == 0:
            n_reg[:] = 0
        name = "rnd%d" % ci
        r = run_st(s_end, down, up, n_reg, cur, lim)
        out.update(prefixed(name + "_", dict(s_end=np.array(s_end), down=down, up=up, n_reg=n_reg,
                                             cur_vel=np.array(cur), lim=np.array(lim), **r)))
        names.append(name)
    out["cases"] = np.array(names)
    save("st.npz", **out)


# ---------------------------------------------------------------- pipeline
def gen_pipe():
    rng = np.random.default_rng(41)
    out = {}
    names = []
    scenes = []
    # scene 0: straight line, KAT-1 obstacles + one dynamic obstacle crossing ahead
    scenes.append(dict(c=0.0, n_knots=7, x0=-2.0, vel=(5.0, 0.0),
                       st=[((8, 0.2), 1.0), ((14, -1.0), 1.0)], dy=[((15.0, 3.0), (0.5, -0.8), 0.9)]))
    # scene 1: curved, sparse obstacles, found
    scenes.append(dict(c=0.004, n_knots=12, x0=-2.0, vel=(6.0, 0.3),
                       st=[((12, 1.4), 0.6), ((30, -0.5), 0.6)], dy=[((20.0, 0.6), (2.0, 0.0), 0.8)]))
    # scene 2: straight, static walls plus a moving wall: more than three blocked layers, not found
    st2 = [((float(x), float(y)), 0.9) for x in (5.0, 12.0, 19.0) for y in (-2.0, -1.0, 0.0, 1.0, 2.0)]
    dy2 = [((19.5, float(y)), (2.5, 0.0), 0.9) for y in (-2.0, -1.0, 0.0, 1.0, 2.0)]
    scenes.append(dict(c=0.0, n_knots=10, x0=-3.1, vel=(4.0, -0.4), st=st2, dy=dy2))
    # scene 4: curved, obstacle dead ahead in the first edge and one later, dynamic crossing
    scenes.append(dict(c=0.003, n_knots=9, x0=-2.0, vel=(5.5, 0.2),
                       st=[((5.0, 0.1), 0.7), ((13.0, 0.9), 0.7), ((18.0, -0.8), 0.7)],
                       dy=[((9.0, -3.0), (0.0, 0.9), 0.8), ((25.0, 0.3), (-3.0, 0.0), 0.8)]))
    # scene 3: long curved line, no obstacles
    scenes.append(dict(c=0.002, n_knots=20, x0=-2.0, vel=(7.5, 0.0), st=[], dy=[]))
    for si, sc in enumerate(scenes):
        name = "p%d" % si
        rm = RobotMap(np.eye(2), np.zeros(2))
        rm.add_robot(np.zeros(2), np.array(sc["vel"]), 0.0)
        xs = np.array([sc["x0"] + 4.5 * i for i in range(sc["n_knots"])])
        knots = [np.array([x, sc["c"] * x * x]) for x in xs]
        rm.add_ref_line(knots, 3.5, 5, len(knots), 0)
        for pos, rad in sc["st"]:
            quiet(rm.add_obstacle, np.array(pos, dtype=float), np.zeros(2), rad)
        for pos, vel, rad in sc["dy"]:
            quiet(rm.add_obstacle, np.array(pos, dtype=float), np.array(vel, dtype=float), rad)
        d = dict(rx=np.array(rm.rx_list), ry=np.array(rm.ry_list), ego_vel=np.array(sc["vel"]),
                 st_pos=np.array([o.ob_pos for o in rm.static_ob]).reshape(-1, 2),
                 st_rad=np.array([o.ob_dist for o in rm.static_ob]),
                 dy_pos=np.array([o.ob_pos_list for o in rm.dynamic_ob]).reshape(-1, 11, 2),
                 dy_rad=np.array([o.ob_dist for o in rm.dynamic_ob]))
        sl = quiet(SLMap, rm)
        d.update(s_map=sl.s_map, l_map=sl.l_map, ego=sl.ego_point, robot_vel=np.array(sl.robot_vel),
                 ob=np.array(sl.ob_list).reshape(-1, 2), ob_dist=np.array(sl.ob_dist),
                 st_traj=np.array([o.ob_pos for o in sl.st_ob_traj]).reshape(-1, 2),
                 dy_traj=np.array([o.ob_pos_list for o in sl.dy_ob_traj]).reshape(-1, 11, 2))
        pp = PathPlanner(sl)
        found, curve = quiet(pp.plan)
        d.update(found=np.array(found), cost_map=pp.cost_map, index_map=pp.index_map.astype(np.int32))
        print(name, "found", found, pp.path_ind_list)
        if found:
            d["path_idx"] = np.array(pp.path_ind_list, dtype=np.int32)
            pb = sl.path_sampling(curve)
            d["path_sl"] = np.array(pb)
            st = STMap(sl.converter, pb[-1][0], 5)
            st.add_obstacle(pb, sl.st_ob_traj, sl.dy_ob_traj)
            rmax = max(1, max(len(o.s_down) for o in st.ob_mat))
            down = np.zeros((st.n_t, rmax))
            up = np.zeros((st.n_t, rmax))
            n_reg = np.zeros(st.n_t, dtype=np.int32)
            for i, o in enumerate(st.ob_mat):
                n_reg[i] = len(o.s_down)
                down[i, :n_reg[i]] = o.s_down
                up[i, :n_reg[i]] = o.s_up
            sp = SpeedPlanner(st, sl.robot_vel, 25.0 * 5.0 / 18.0)
            ss, vv = quiet(sp.plan)
            d.update(st_n_s=np.array(st.n_s), st_n_t=np.array(st.n_t), st_down=down, st_up=up, st_n_reg=n_reg,
                     st_cost=np.array([[st.map[i][j].cost for j in range(st.n_s)] for i in range(st.n_t)]),
                     st_pre=np.array([[st.map[i][j].pre_ind for j in range(st.n_s)] for i in range(st.n_t)],
                                     dtype=np.int32),
                     path_s_ind=np.array(sp.path_s_ind, dtype=np.int32), ss=ss, vv=vv)
            pb2 = sl.path_sampling(curve, False, ss)
            d["path_sl2"] = np.array(pb2)
            d["path_xy"] = np.array(quiet(sl.path_convert, pb2))
        out.update(prefixed(name + "_", d))
        names.append(name)
    out["cases"] = np.array(names)
    save("pipe.npz", **out)


# ---------------------------------------------------------------- world-frame scenes
def gen_world():
    """Scenes given in the world frame, run through the sequence of
    PlannerInterface.run_step (coor_trans, path_plan, speed_plan, both
    path_convert calls) with the reference's classes; CARLA objects replaced by
    plain arrays.  Both path planners are recorded."""
    import math
    rng = np.random.default_rng(77)
    out = {}
    names = []
    designs = [
        # curvature, knots, first knot x (robot frame), ego pose (x, y, theta), ego speed (robot frame)
        dict(c=0.0, nk=7, x0=-2.0, pose=(103.5, -41.25, 0.6), vel=(5.0, 0.0)),
        dict(c=0.004, nk=12, x0=-2.0, pose=(-220.0, 87.5, 2.4), vel=(6.0, 0.3)),
        dict(c=-0.003, nk=10, x0=-1.5, pose=(12.0, 300.0, -1.9), vel=(4.0, -0.2)),
        dict(c=0.002, nk=16, x0=-2.5, pose=(-5.0, -7.0, 3.1), vel=(7.0, 0.1)),
        dict(c=0.0, nk=9, x0=-2.0, pose=(55.0, 66.0, -0.3), vel=(3.0, 0.0)),
        dict(c=0.001, nk=8, x0=-2.0, pose=(0.0, 0.0, 0.0), vel=(5.0, 0.0)),
    ]
    for si, sc in enumerate(designs):
        name = "w%d" % si
        ex, ey, th = sc["pose"]
        ego_rot = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
        ego_pos = np.array([ex, ey])
        to_world = lambda p: ego_rot.dot(np.asarray(p, dtype=float)) + ego_pos  # noqa: E731
        ego_vel = ego_rot.dot(np.array(sc["vel"], dtype=float))
        xs = np.array([sc["x0"] + 4.5 * i for i in range(sc["nk"])])
        ref_line = [to_world([x, sc["c"] * x * x]) for x in xs]
        # obstacles: robot-frame design -> world; classes mixed on purpose
        obs = []
        n_ob = [6, 9, 12, 4, 0, 7][si]
        for k in range(n_ob):
            kind = k % 6
            if kind == 0:    # static, near, around the lane
                pr, vr = (rng.uniform(4, 18), rng.uniform(-2.5, 2.5)), (0.0, 0.0)
            elif kind == 1:  # dynamic, near
                pr, vr = (rng.uniform(6, 19), rng.uniform(-3, 3)), (rng.uniform(-2, 2), rng.uniform(-1, 1))
            elif kind == 2:  # behind the car
                pr, vr = (rng.uniform(-9, -1.2), rng.uniform(-2, 2)), (0.0, 0.0) if k % 2 else (1.0, 0.0)
            elif kind == 3:  # static, out of range
                pr, vr = (rng.uniform(21, 40), rng.uniform(-3, 3)), (0.0, 0.0)
            elif kind == 4:  # dynamic, starts far and approaches
                pr, vr = (rng.uniform(22, 30), rng.uniform(-1, 1)), (rng.uniform(-4, -2), 0.0)
            else:            # dynamic, far and leaving
                pr, vr = (rng.uniform(21, 30), rng.uniform(-1, 1)), (rng.uniform(1, 3), 0.0)
            obs.append((to_world(pr), ego_rot.dot(np.array(vr, dtype=float)), float(rng.uniform(0.5, 1.0))))
        d = dict(ego_pos=ego_pos, ego_theta=np.array(th), ego_rot=ego_rot, ego_vel=ego_vel,
                 ref_line=np.array(ref_line),
                 ob_pos=np.array([o[0] for o in obs]).reshape(-1, 2), ob_vel=np.array([o[1] for o in obs]).reshape(-1, 2),
                 ob_rad=np.array([o[2] for o in obs]))
        # coor_trans (planner_interface.py:98-131)
        rm = RobotMap(ego_rot, ego_pos)
        rm.add_robot(ego_pos, ego_vel, 0.0)
        rm.add_ref_line(ref_line, 3.5, 5, len(ref_line), 0)
        near = [bool(quiet(rm.add_obstacle, o[0], o[1], o[2])) for o in obs]
        d.update(near=np.array(near, dtype=bool), rx=np.array(rm.rx_list), ry=np.array(rm.ry_list),
                 robot_vel_xy=np.array(rm.robot.vel),
                 st_pos=np.array([o.ob_pos for o in rm.static_ob]).reshape(-1, 2),
                 st_rad=np.array([o.ob_dist for o in rm.static_ob]),
                 dy_pos=np.array([o.ob_pos_list for o in rm.dynamic_ob]).reshape(-1, 11, 2),
                 dy_vel=np.array([o.ob_vel for o in rm.dynamic_ob]).reshape(-1, 2),
                 dy_rad=np.array([o.ob_dist for o in rm.dynamic_ob]))
        sl = quiet(SLMap, rm)
        for tag, cls in (("track", TrackPathPlanner), ("dp", PathPlanner)):
            found, curve = quiet(cls(sl).plan)
            d[tag + "_found"] = np.array(bool(found))
            if not found:
                continue
            # speed_plan (planner_interface.py:140-159)
            pb = sl.path_sampling(curve)
            st = STMap(sl.converter, pb[-1][0], 5)
            st.add_obstacle(pb, sl.st_ob_traj, sl.dy_ob_traj)
            ss, speed_buff = quiet(SpeedPlanner(st, sl.robot_vel, 25.0 * 5.0 / 18.0).plan)
            pb2 = sl.path_sampling(curve, False, ss)
            robot_xy = quiet(sl.path_convert, pb2)
            world_xy = rm.path_convert(robot_xy)
            d.update({tag + "_ss": np.array(ss), tag + "_speed": np.array(speed_buff),
                      tag + "_path_sl": np.array(pb2), tag + "_robot_xy": np.array(robot_xy),
                      tag + "_world_xy": np.array(world_xy)})
        print(name, "near", near, "track", bool(d["track_found"]), "dp", bool(d["dp_found"]))
        out.update(prefixed(name + "_", d))
        names.append(name)
    out["cases"] = np.array(names)
    save("world.npz", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (gen_primitives, gen_c2f, gen_sl, gen_st, gen_pipe, gen_world):
        if not only or fn.__name__[4:] in only:
            fn()
