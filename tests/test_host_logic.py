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
