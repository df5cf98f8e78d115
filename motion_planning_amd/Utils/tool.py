"""Numeric helpers with the semantics of the reference's Utils/tool.py
(the CARLA-bound helpers of that module are out of scope here)."""
import math

import numpy as np


def get_arange(t_st, t_en, dt):
    """Samples from t_st to t_en INCLUSIVE (reference Utils/tool.py:27-33).

    Empty when the span is shorter than dt; otherwise the np.arange grid with
    t_en appended unless the grid already ends on it, so a span of n*dt has
    n+1 samples."""
    if t_st + dt > t_en:
        return np.array([])
    grid = np.arange(t_st, t_en, dt)
    return grid if grid[-1] == t_en else np.append(grid, t_en)


def cal_dist(p1, p2):
    """Euclidean distance of two points (reference Utils/tool.py:58-59)."""
    return math.sqrt((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2)


def cal_dist_arr(arr, p):
    """Distance from p to every column of arr[2, K]; returns (minimum, point of
    the FIRST minimum) (reference Utils/tool.py:62-72)."""
    p = np.asarray(p, dtype=np.float64).reshape(2, 1)
    diff = np.asarray(arr, dtype=np.float64) - p
    dist = np.sqrt(diff[0] ** 2 + diff[1] ** 2)
    k = int(np.argmin(dist))
    return dist[k], np.array([arr[0][k], arr[1][k]])


def to_point(x, y):
    return np.array([x, y])


def check_static(vel):
    """An obstacle slower than 1 mm/s (L1 norm) is static (reference Utils/tool.py:41-42)."""
    return np.sum(np.abs(vel)) < 0.001
