"""Host-side pieces of the drop-in API (no GPU): spline / arange helpers against
the reference's golden outputs, obstacle rollout, robot-frame map filters."""
import numpy as np

from conftest import load_golden

from motion_planning_amd.Model.curve import Curve, QuinticPoly
from motion_planning_amd.Model.obstacle import DynamicObstacle
from motion_planning_amd.Planning.Interface.robot_map import RobotMap
from motion_planning_amd.Utils.tool import cal_dist_arr, get_arange


def close(a, b):
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)


def test_get_arange_golden():
    g = load_golden("primitives.npz")
    off = 0
    for (a, b, dt), n in zip(g["ar_in"], g["ar_len"]):
        v = get_arange(a, b, dt)
        assert np.array_equal(v, g["ar_val"][off:off + n])
        off += n
    assert len(get_arange(0.0, 0.2, 0.5)) == 0
    assert len(get_arange(0.0, 4.5, 0.5)) == 10          # both ends counted


def test_quintic_golden():
    g = load_golden("primitives.npz")
    for r, coef, ev in zip(g["q_in"], g["q_coef"], g["q_eval"]):
        q = QuinticPoly(*r)
        assert np.array_equal([q.a0, q.a1, q.a2, q.a3, q.a4, q.a5], coef)
        for o in range(4):
            assert np.array_equal(q.calc_point(g["q_t"], o), ev[o])
    # evaluated outside [0, T] as the SL edge costs do
    assert QuinticPoly(0, 0, 0, 1, 0, 0, 8).calc_point(16, 0) == 32


def test_curve_golden():
    g = load_golden("primitives.npz")
    for ci in range(int(g["n_curves"])):
        p = "cv%d_" % ci
        dt, vec = g[p + "dtvec"]
        cv = Curve(g[p + "t"], dt, g[p + "x"], vec)
        tt = g[p + "tt"]
        for o in range(4):
            assert np.array_equal(cv.calc_point_arr(tt, o), g[p + "arr"][o])
        for o in range(3):
            assert np.array_equal([cv.calc_point(u, o) for u in tt], g[p + "scalar"][o])
        assert np.array_equal([cv.calc_arc_len(cv.t_bios, u) for u in tt], g[p + "arc"])
        for (a, b), ref in zip(g[p + "proj_in"], g[p + "proj_out"]):
            d, q = cv.projection(a, b)
            assert np.array_equal([d, q[0], q[1]], ref)


def test_curve_array_form_lags_behind_scalar_form():
    cv = Curve(np.array([0.0, 2, 4, 6]), 0.5, np.array([0.0, 1, 0, 2]), 0.0)
    tt = get_arange(0, 6, 0.5)
    np.testing.assert_allclose(cv.calc_point_arr(tt, 0)[5], 1.23596191, rtol=1e-8)
    np.testing.assert_allclose(cv.calc_point(2.5, 0), 0.8807373, rtol=1e-7)
    # straight 13.5 m line of three 4.5 m segments measures 15.0 (one extra sample per segment)
    line = Curve(np.array([0.0, 4.5, 9.0, 13.5]), 0.5, np.zeros(4), 0.0)
    assert line.calc_arc_len(0.0, 13.5) == 15.0
    assert line.calc_arc_len(0.0, 0.2) == 0.0


def test_cal_dist_arr_first_minimum():
    arr = np.array([[0.0, 1.0, 2.0, 1.0], [0.0, 1.0, 0.0, -1.0]])
    d, p = cal_dist_arr(arr, np.array([1.0, 0.0]))
    assert d == 1.0 and tuple(p) == (0.0, 0.0)


def test_dynamic_rollout_and_robot_map_filters():
    ob = DynamicObstacle(np.array([1.0, 2.0]), np.array([2.0, -1.0]), 0.8)
    assert len(ob.ob_pos_list) == 11
    close(ob.ob_pos_list[-1], [11.0, -3.0])
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    rm = RobotMap(R, np.array([5.0, -2.0]))
    rm.add_robot(np.array([5.0, -2.0]), R @ np.array([4.0, 0.0]), 0.0)
    close(rm.robot.pos, [0, 0])
    close(rm.robot.vel, [4, 0])
    to_world = rm.WRC.robot_to_world
    assert rm.add_obstacle(to_world([8.0, 1.0]), np.zeros(2), 1.0)            # static, near
    assert not rm.add_obstacle(to_world([-3.0, 0.0]), np.zeros(2), 1.0)       # behind
    assert not rm.add_obstacle(to_world([25.0, 0.0]), np.zeros(2), 1.0)       # too far
    assert rm.add_obstacle(to_world([24.0, 0.0]), R @ np.array([-3.0, 0.0]), 1.0)  # approaches within 20 m
    assert len(rm.static_ob) == 1 and len(rm.dynamic_ob) == 1
    close(rm.path_convert([np.array([1.0, 0.0])])[0], to_world([1.0, 0.0]))


def _local_control_scalar(path, speed, loc, local_ind, min_distance=3.0, replan_ind=10):
    """Scalar restatement of the target selection in LocalPlanner.local_contorl
    (reference Planning/Interface/local_planner.py:284-311): returns
    (tmp_ind, local_ind, re_plan)."""
    import math
    dist = lambda p, q: math.sqrt((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2)  # noqa: E731
    re_plan = False
    if len(speed) == 0 or max(speed) < 0.1 or dist(path[0], path[-1]) < 1.0:
        tmp_ind = -1
    elif len(path) == 0:
        tmp_ind = -1
    elif local_ind < len(path) - replan_ind:
        tmp_ind = local_ind
        for i in range(local_ind, len(path)):
            if dist(path[i], loc) < min_distance:
                tmp_ind = i
        local_ind = tmp_ind
    else:
        tmp_ind = len(path) - replan_ind
        re_plan = True
    return tmp_ind, local_ind, re_plan


def test_select_targets_matches_local_control():
    from motion_planning_amd.Planning.Interface.planner_interface import select_targets
    rng = np.random.default_rng(3)
    B, P = 64, 60
    n_path = rng.integers(0, P + 1, B)
    n_path[:4] = [0, 5, 10, 11]
    px, py, sp = np.zeros((B, P)), np.zeros((B, P)), np.zeros((B, P))
    veh = np.zeros((B, 2))
    local = np.zeros(B, np.int64)
    for b in range(B):
        n = n_path[b]
        step = rng.uniform(0.005, 1.2)
        px[b, :n] = step * np.arange(n) + rng.uniform(-50, 50)
        py[b, :n] = 0.1 * np.sin(np.arange(n)) + rng.uniform(-50, 50)
        sp[b, :n] = rng.uniform(0, 7, n) * (rng.random() > 0.15)
        if n:
            k = rng.integers(0, n)
            veh[b] = [px[b, k] + rng.uniform(-2, 2), py[b, k] + rng.uniform(-2, 2)]
            local[b] = rng.integers(0, n)
    r = select_targets(px, py, n_path, sp, veh, local)
    for b in range(B):
        n = n_path[b]
        path = [(px[b, i], py[b, i]) for i in range(n)]
        tmp, loc_new, re_plan = _local_control_scalar(path, list(sp[b, :n]), veh[b], int(local[b]))
        assert r["ind"][b] == (tmp if tmp >= 0 else -1), b
        assert r["local_ind"][b] == loc_new, b
        assert bool(r["replan"][b]) == re_plan, b
        if tmp >= 0:
            assert np.array_equal(r["target_xy"][b], path[tmp])
            assert r["target_speed_kmh"][b] == sp[b, tmp] * 18.0 / 5.0
        else:
            assert np.all(np.isnan(r["target_xy"][b])) and r["target_speed_kmh"][b] == 0.0
