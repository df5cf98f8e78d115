"""Obstacle containers with the reference's field names (Model/obstacle.py).
All obstacles are circles: a centre and a radius."""
from ..Utils.tool import cal_dist


class VehicleBox:
    def __init__(self, pos, vel, radius):
        self.pos, self.vel, self.radius = pos, vel, radius

    def cal_dist(self, point):
        return cal_dist(self.pos, point)


class StaticObstacle:
    def __init__(self, ob_pos, ob_vel, ob_dist):
        self.ob_pos, self.ob_vel, self.ob_dist = ob_pos, ob_vel, ob_dist

    def cal_dist(self, point):
        return cal_dist(self.ob_pos, point)


class ObstacleTrajectory:
    """Positions of one moving obstacle at t = 0, 0.5, ... in some frame."""

    def __init__(self, ob_dist):
        self.ob_pos_list = []
        self.ob_dist = ob_dist
        self.dt = 0.5


class DynamicObstacle:
    """Constant-velocity rollout over eval_time seconds in dt steps: 11 frames
    for the defaults (reference Model/obstacle.py:36-53)."""

    def __init__(self, ob_pos, ob_vel, ob_dist):
        self.ob_vel = ob_vel
        self.ob_dist = ob_dist
        self.eval_time = 5
        self.dt = 0.5
        self.ob_pos_list = [ob_pos]
        t = self.dt
        while t <= self.eval_time:
            self.ob_pos_list.append(self.ob_pos_list[-1] + self.ob_vel * self.dt)
            t += self.dt

    def cal_min_dist(self, point):
        return min(cal_dist(p, point) for p in self.ob_pos_list)

    def cal_dist(self, point, t):
        if t > self.eval_time:
            print("t is out of prediction range.")
            return 0.0
        return cal_dist(self.ob_pos_list[int(1.0 * t / self.dt)], point)
