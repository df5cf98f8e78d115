"""CARLA-free planner front end: the reference's PlannerInterface.run_step
(Planning/Interface/planner_interface.py:35-95) for many scenes per call, and
the piece of LocalPlanner.local_contorl (Planning/Interface/local_planner.py:
284-311) that turns a plan into the PID controller's target.

The reference reads its inputs from CARLA actors; here a scene is plain data
(`Scene`).  run_step returns, per scene, what the reference returns:
(path_buff, speed_buff) with path_buff a list of `Location` (x, y attributes,
standing in for carla.Location) and speed_buff the speeds in m/s, or ([], [])
when no path was found.  All planning stages run on the GPU through
pipeline.plan_world_batch; there is no CPU fallback.
"""
import math
from collections import namedtuple

import numpy as np

from ... import pipeline

Location = namedtuple("Location", ["x", "y"])


class Scene:
    """What PlannerInterface.coor_trans reads from the simulator for one replan
    (planner_interface.py:98-131): ego pose and velocity, the waypoints ahead,
    and every other actor as (position, velocity, bounding radius)."""

    def __init__(self, ego_pos, ego_vel, ego_theta, waypoints, obstacles=(), change_lane=False):
        self.ego_pos = np.asarray(ego_pos, dtype=np.float64).reshape(2)
        self.ego_vel = np.asarray(ego_vel, dtype=np.float64).reshape(2)
        self.ego_theta = float(ego_theta)
        self.waypoints = np.asarray(waypoints, dtype=np.float64).reshape(-1, 2)
        self.obstacles = [(np.asarray(p, dtype=np.float64).reshape(2), np.asarray(v, dtype=np.float64).reshape(2),
                           float(r)) for p, v, r in obstacles]
        # the reference computes cal_theta_ind from the command but RobotMap.add_ref_line
        # ignores it (robot_map.py:31-44), so a lane change plans like lane following
        self.change_lane = bool(change_lane)

    def rotation(self):
        """ego_rot of coor_trans (planner_interface.py:112-113)."""
        c, s = math.cos(self.ego_theta), math.sin(self.ego_theta)
        return np.array([[c, -s], [s, c]])


class PlannerInterface:
    """Batched stand-in for the reference class of the same name.

    path_planner "track" is what the reference instantiates (TrackPathPlanner,
    planner_interface.py:136); "dp" runs the SL lattice DP (PathPlanner) instead."""

    l_width = 3.5   # lane width, planner_interface.py:99
    n_l = 5         # lateral samples, planner_interface.py:100
    plan_time = 5   # planner_interface.py:144
    speed_lim = 25.0 * 5.0 / 18.0   # planner_interface.py:151

    def __init__(self, scenes, path_planner="track"):
        self.scenes = list(scenes)
        self.path_planner = path_planner
        self.result = None
        self.status = None

    def pack(self):
        """Scenes as padded arrays (the layout pipeline.plan_world_batch takes)."""
        B = len(self.scenes)
        W = max(len(s.waypoints) for s in self.scenes)
        O = max([len(s.obstacles) for s in self.scenes] + [0])
        a = dict(rot=np.zeros((B, 2, 2)), trans=np.zeros((B, 2)), ego_vel_w=np.zeros((B, 2)),
                 way_xy=np.zeros((B, W, 2)), n_way=np.zeros(B, np.int32), ob_pos=np.zeros((B, O, 2)),
                 ob_vel=np.zeros((B, O, 2)), ob_rad=np.zeros((B, O)), n_ob=np.zeros(B, np.int32))
        for b, s in enumerate(self.scenes):
            a["rot"][b], a["trans"][b], a["ego_vel_w"][b] = s.rotation(), s.ego_pos, s.ego_vel
            a["n_way"][b] = len(s.waypoints)
            a["way_xy"][b, :len(s.waypoints)] = s.waypoints
            a["n_ob"][b] = len(s.obstacles)
            for k, (p, v, r) in enumerate(s.obstacles):
                a["ob_pos"][b, k], a["ob_vel"][b, k], a["ob_rad"][b, k] = p, v, r
        return a

    def run_step(self):
        """Plans every scene; returns [(path_buff, speed_buff), ...] in scene order."""
        if not self.scenes:
            return []
        a = self.pack()
        res = pipeline.plan_world_batch(a["rot"], a["trans"], a["ego_vel_w"], a["way_xy"], a["n_way"], a["ob_pos"],
                                        a["ob_vel"], a["ob_rad"], a["n_ob"], l_width=self.l_width, n_l=self.n_l,
                                        speed_lim=self.speed_lim, plan_time=self.plan_time,
                                        path_planner=self.path_planner)
        self.result = res
        self.status = res["status"].cpu().numpy()
        out = [([], []) for _ in self.scenes]
        wx, wy = res["world_x"].cpu().numpy(), res["world_y"].cpu().numpy()
        # PIPE_OK guarantees n_final == n_prof > 0: a scene whose result could not be sampled
        # carries PIPE_INDEX / PIPE_ST_CAPACITY and yields ([], []) like an unplanned one
        vv, n_final = res["vv"].cpu().numpy(), res["n_final"].cpu().numpy()
        for b in np.flatnonzero(self.status == pipeline.PIPE_OK):
            n = int(n_final[b])
            out[b] = ([Location(float(wx[b, i]), float(wy[b, i])) for i in range(n)], [float(v) for v in vv[b, :n]])
        return out


def select_targets(path_x, path_y, n_path, speed, vehicle_xy, local_ind, min_distance=3.0, replan_ind=10):
    """Target point and speed for the PID controller, for B vehicles at once:
    the selection LocalPlanner.local_contorl makes before it calls
    VehiclePIDController.run_step(target_speed_kmh, target_waypoint)
    (local_planner.py:284-311, 327).

    path_x / path_y / speed [B, P] with n_path[b] valid entries (0 = no plan),
    vehicle_xy [B, 2], local_ind [B] (the index kept from the previous control
    step); min_distance and replan_ind default to the reference's 3 m and 10
    (local_planner.py:76, 130).  Returns dict(ind [B] (-1 = stop), target_xy [B, 2], target_speed_kmh
    [B], local_ind [B] (updated), replan [B] bool)."""
    path_x, path_y, speed = (np.asarray(v, dtype=np.float64) for v in (path_x, path_y, speed))
    n_path = np.asarray(n_path, dtype=np.int64)
    vehicle_xy = np.asarray(vehicle_xy, dtype=np.float64).reshape(-1, 2)
    local_ind = np.asarray(local_ind, dtype=np.int64).copy()
    B, P = path_x.shape
    col = np.arange(P)[None, :]
    valid = col < n_path[:, None]
    last = np.clip(n_path - 1, 0, None)
    rows = np.arange(B)
    # "No speed": empty plan, every speed below 0.1 m/s, or a path shorter than 1 m
    vmax = np.where(valid, speed, -np.inf).max(axis=1, initial=-np.inf)
    span = np.sqrt((path_x[rows, 0] - path_x[rows, last]) ** 2 + (path_y[rows, 0] - path_y[rows, last]) ** 2)
    stop = (n_path == 0) | (vmax < 0.1) | (span < 1.0)
    # furthest path point from local_ind on that lies within min_distance of the vehicle
    dist = np.sqrt((path_x - vehicle_xy[:, :1]) ** 2 + (path_y - vehicle_xy[:, 1:]) ** 2)
    cand = valid & (col >= local_ind[:, None]) & (dist < min_distance)
    reached = np.where(cand.any(axis=1), P - 1 - np.argmax(cand[:, ::-1], axis=1), local_ind)
    near_end = ~(local_ind < n_path - replan_ind)          # "Path end": hold the replan point
    ind = np.where(near_end, n_path - replan_ind, reached)
    new_local = np.where(stop | near_end, local_ind, reached)
    ind = np.where(stop | (ind < 0), -1, ind)   # a plan shorter than replan_ind points also stops
    ok = ind >= 0
    safe = np.clip(ind, 0, P - 1)
    target_xy = np.where(ok[:, None], np.stack([path_x[rows, safe], path_y[rows, safe]], 1), np.nan)
    target_speed = np.where(ok, speed[rows, safe] * 18.0 / 5.0, 0.0)   # speed_conversion(v, "m"): m/s -> km/h
    return dict(ind=ind, target_xy=target_xy, target_speed_kmh=target_speed, local_ind=new_local,
                replan=~stop & near_end)
