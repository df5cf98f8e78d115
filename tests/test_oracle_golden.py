"""The CPU oracle against golden vectors produced by running the reference.

Fixtures: tests/golden/*.npz, written by tools/gen_golden.py (which imports the
reference in the build container).  Tolerances: indices exact; floating point
1e-12 relative (the reference's NumPy may use SIMD pow, so the last bits differ).
"""
import numpy as np
import pytest

from conftest import load_golden

RTOL = 1e-12


def close(a, b, rtol=RTOL, atol=1e-13):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_np_sum_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    for n in [0, 1, 3, 7, 8, 9, 17, 64, 127, 128, 129, 200, 1000]:
        a = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, n)
        assert oracle.np_sum(a) == float(np.sum(a))


def test_get_arange(oracle):
    g = load_golden("primitives.npz")
    off = 0
    for (a, b, dt), n in zip(g["ar_in"], g["ar_len"]):
        v = oracle.get_arange(a, b, dt)
        assert len(v) == n
        assert np.array_equal(v, g["ar_val"][off:off + n])
        off += n


def test_quintic(oracle):
    g = load_golden("primitives.npz")
    for r, coef, ev in zip(g["q_in"], g["q_coef"], g["q_eval"]):
        q = oracle.QuinticPoly(*r)
        assert np.array_equal(q.a, coef)
        for o in range(4):
            close(q.calc_point(g["q_t"], o), ev[o])


def test_curve(oracle):
    g = load_golden("primitives.npz")
    for ci in range(int(g["n_curves"])):
        p = "cv%d_" % ci
        dt, vec = g[p + "dtvec"]
        cv = oracle.Curve(g[p + "t"], dt, g[p + "x"], vec)
        tt = g[p + "tt"]
        tx, xx = cv.samples()
        assert np.array_equal(tx, tt)
        for o in range(4):
            close(cv.calc_point_arr(tt, o), g[p + "arr"][o])
        for o in range(3):
            close([cv.calc_point(u, o) for u in tt], g[p + "scalar"][o])
        close([cv.calc_arc_len(cv.t_bios, u) for u in tt], g[p + "arc"])
        close([cv.calc_arc_len(tt[len(tt) // 3], u) for u in tt], g[p + "arc_mid"])
        for (a, b), ref in zip(g[p + "proj_in"], g[p + "proj_out"]):
            d, q = cv.projection(a, b)
            close([d, q[0], q[1]], ref)


def test_curve_lag_known_answer(oracle):
    # just after a knot the array form keeps the previous segment
    cv = oracle.Curve(np.array([0.0, 2, 4, 6]), 0.5, np.array([0.0, 1, 0, 2]), 0.0)
    tt = oracle.get_arange(0, 6, 0.5)
    arr = cv.calc_point_arr(tt, 0)
    close(arr[5], 1.23596191, rtol=1e-8)
    close(cv.calc_point(2.5, 0), 0.8807373, rtol=1e-7)


def test_c2f(oracle):
    g = load_golden("c2f.npz")
    for li in range(int(g["n_lines"])):
        p = "l%d_" % li
        conv = oracle.CartesianFrenetConverter(0.0, 0.0, g[p + "rx"], g[p + "ry"], 0.0)
        close([conv.rx_ori, conv.ry_ori, conv.rx_max], g[p + "ori"])
        o0 = np.array([conv.cartesian_to_frenet(x, y) for x, y in g[p + "pts"]])
        close(o0, g[p + "o0"])
        o1 = np.array([conv.cartesian_to_frenet(x, y, vx, vy, 1)
                       for (x, y), (vx, vy) in zip(g[p + "pts"], g[p + "vel"])])
        close(o1, g[p + "o1"])
        tx, xx, S, th, kr = conv.tables()
        assert np.array_equal(tx, g[p + "tab_x"])
        close(xx, g[p + "tab_y"])
        close(S, g[p + "tab_S"])
        close(th, g[p + "tab_th"])
        close(kr, g[p + "tab_kr"])
        f = np.array([conv.frenet_to_cartesian(a, b) for a, b in g[p + "f2c_in"]])
        close(f, g[p + "f2c_out"])


def sl_cases():
    return [str(c) for c in load_golden("sl.npz")["cases"]]


@pytest.mark.parametrize("case", sl_cases())
def test_sl_find_path(oracle, case):
    g = load_golden("sl.npz")
    p = case + "_"
    want_edges = (p + "edge_cost") in g.files
    r = oracle.sl_find_path(g[p + "s_map"], g[p + "l_map"], int(g[p + "mid_l"]), g[p + "ego"], g[p + "ob"],
                            float(g[p + "ob_dist"]), want_edges=want_edges)
    assert r["found"] == bool(g[p + "found"])
    close(r["cost_map"], g[p + "cost_map"])
    assert np.array_equal(r["index_map"], g[p + "index_map"])
    if r["found"]:
        assert np.array_equal(r["path_idx"], g[p + "path_idx"])
    if want_edges:
        e, ge = r["edge_cost"], g[p + "edge_cost"]
        done = r["layers_done"]
        close(e[:done], ge[:done])


def test_sl_kat1_known_answer(oracle):
    # cfg1: straight line, 3 static obstacles, values recorded from the reference
    g = load_golden("sl.npz")
    assert np.array_equal(g["kat1_s_map"][:, 0], [8, 16, 28])
    assert np.array_equal(g["kat1_path_idx"], [4, 4, 3])
    r = oracle.sl_find_path(g["kat1_s_map"], g["kat1_l_map"], 2, g["kat1_ego"], g["kat1_ob"], 1.0)
    close(r["cost_map"][2], [30489.997146197715, 31684.922113717363, 7102.9289631603351,
                             5259.5870597907406, 30025.887995570331])
    assert np.array_equal(r["index_map"], [[-1] * 5, [1, 3, 2, 3, 4], [0, 1, 1, 4, 4]])


def test_sl_no_early_exit_keeps_prefix(oracle):
    g = load_golden("sl.npz")
    p = "wall_nf_"
    a = oracle.sl_find_path(g[p + "s_map"], g[p + "l_map"], int(g[p + "mid_l"]), g[p + "ego"], g[p + "ob"],
                            float(g[p + "ob_dist"]))
    b = oracle.sl_find_path(g[p + "s_map"], g[p + "l_map"], int(g[p + "mid_l"]), g[p + "ego"], g[p + "ob"],
                            float(g[p + "ob_dist"]), no_early_exit=True)
    assert not a["found"] and not b["found"]
    d = a["layers_done"]
    assert d < b["layers_done"] == g[p + "s_map"].shape[0]
    assert np.array_equal(a["cost_map"][:d], b["cost_map"][:d])


def test_sl_isolated_edges(oracle):
    g = load_golden("sl.npz")
    for (p1s, p1l, p2s, p2l), want in zip(g["edge_in"], g["edge_cost"]):
        s0 = p1s + 1e-9 if p1s == 0 else p1s
        s_map = np.array([[p1s] * 3, [p2s] * 3])
        l_map = np.array([[p1l, p2l, 0.0], [p1l, p2l, 0.0]])
        del s0
        c, parts = oracle.sl_cal_cost(s_map, l_map, 2, np.array([0.0, p1l]), g["edge_ob"],
                                      float(g["edge_ob_dist"]), 0, 0, 1, 1)
        close(c, want)


def st_cases():
    return [str(c) for c in load_golden("st.npz")["cases"]]


@pytest.mark.parametrize("case", st_cases())
def test_st_find_speed(oracle, case):
    g = load_golden("st.npz")
    p = case + "_"
    n_s, n_t = oracle.st_dims(float(g[p + "s_end"]))
    assert (n_s, n_t) == (int(g[p + "n_s"]), int(g[p + "n_t"]))
    r = oracle.st_find_speed(n_t, n_s, g[p + "down"], g[p + "up"], g[p + "n_reg"], float(g[p + "cur_vel"]),
                             float(g[p + "lim"]))
    # the ST arithmetic is restated operation for operation: bit-exact
    assert np.array_equal(r["cost"], g[p + "cost"])
    assert np.array_equal(r["pre"], g[p + "pre"])
    assert np.array_equal(r["path_s_ind"], g[p + "path_s_ind"])
    ss, vv = oracle.speed_profile(r["path_s_ind"], float(g[p + "cur_vel"]))
    close(ss, g[p + "ss"])
    close(vv, g[p + "vv"])


def test_st_kat2_known_answer(oracle):
    g = load_golden("st.npz")
    assert np.array_equal(g["kat2_28_path_s_ind"], [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
    close(g["kat2_28_cost"][10][:3], [750.0, 775.6, 793.2])
    assert len(g["kat2_28_ss"]) == 51 and g["kat2_28_vv"][0] == 5.0


def pipe_cases():
    return [str(c) for c in load_golden("pipe.npz")["cases"]]


@pytest.mark.parametrize("case", pipe_cases())
def test_pipeline(oracle, case):
    g = load_golden("pipe.npz")
    p = case + "_"
    st = [(pos, r) for pos, r in zip(g[p + "st_pos"], g[p + "st_rad"])]
    dy = [(fr, r) for fr, r in zip(g[p + "dy_pos"], g[p + "dy_rad"])]
    sl = oracle.SLMap(g[p + "rx"], g[p + "ry"], 0.0, g[p + "ego_vel"], st, dy)
    close(sl.s_map, g[p + "s_map"])
    close(sl.l_map, g[p + "l_map"])
    close(sl.ego_point, g[p + "ego"])
    close(sl.robot_vel, g[p + "robot_vel"])
    close(np.array(sl.ob_list).reshape(-1, 2), g[p + "ob"])
    assert sl.ob_dist == float(g[p + "ob_dist"])
    close(np.array([q for q, _ in sl.st_ob_traj]).reshape(-1, 2), g[p + "st_traj"])
    close(np.array([q for q, _ in sl.dy_ob_traj]).reshape(-1, 11, 2), g[p + "dy_traj"])
    found, curve, r = oracle.sl_plan(sl)
    assert found == bool(g[p + "found"])
    d = r["layers_done"]
    close(r["cost_map"][:d], g[p + "cost_map"][:d])
    assert np.array_equal(r["index_map"][:d], g[p + "index_map"][:d])
    if not found:
        return
    assert np.array_equal(r["path_idx"], g[p + "path_idx"])
    pb = sl.path_sampling(curve)
    close(pb, g[p + "path_sl"])
    n_s, n_t = oracle.st_dims(pb[-1][0])
    assert (n_s, n_t) == (int(g[p + "st_n_s"]), int(g[p + "st_n_t"]))
    down, up, n_reg = oracle.st_add_obstacle(n_t, pb, sl.st_ob_traj, sl.dy_ob_traj)
    assert np.array_equal(n_reg, g[p + "st_n_reg"])
    rm = g[p + "st_down"].shape[1]
    for i in range(n_t):
        close(down[i, :n_reg[i]], g[p + "st_down"][i, :n_reg[i]])
        close(up[i, :n_reg[i]], g[p + "st_up"][i, :n_reg[i]])
    del rm
    sr = oracle.st_find_speed(n_t, n_s, down, up, n_reg, sl.robot_vel, 25.0 * 5.0 / 18.0)
    close(sr["cost"], g[p + "st_cost"])
    assert np.array_equal(sr["pre"], g[p + "st_pre"])
    assert np.array_equal(sr["path_s_ind"], g[p + "path_s_ind"])
    ss, vv = oracle.speed_profile(sr["path_s_ind"], sl.robot_vel)
    close(ss, g[p + "ss"])
    close(vv, g[p + "vv"])
    pb2 = sl.path_sampling(curve, ss)
    close(pb2, g[p + "path_sl2"])
    close(sl.path_convert(pb2), g[p + "path_xy"])
