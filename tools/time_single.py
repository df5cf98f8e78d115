"""Latency of ONE plan on the GPU (development aid): what a caller that replans a
single vehicle sees per call, launch to synchronised result.

    python3 tools/time_single.py
"""
import sys, time
import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from motion_planning_amd import batch as mb, pipeline as pl

SPEED_LIM = 25.0 * 5.0 / 18.0


def wall(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    # ST speed DP, default sampling, 5 moving blocked intervals, one plan
    base = np.array([(12, 2), (20, 0), (6, 4), (25, -1), (15, 1)], dtype=np.float64)
    for s_end in (28.0, 60.0, 100.0):
        n_s = int(np.floor(s_end / 1.0)) + 1
        s0 = base[:, 0] * (s_end / 28.0)
        down = (s0[None, :] + base[None, :, 1] * 0.5 * np.arange(11)[:, None])[None]
        args = (mb.to_dev(np.array([n_s]), torch.int32), mb.to_dev(down), mb.to_dev(down + 3.0),
                mb.to_dev(np.full((1, 11), 5), torch.int32), mb.to_dev(np.array([5.0])), SPEED_LIM)
        us = wall(lambda: mb.st_dp(*args, want_tables=False, n_s_max=n_s))
        print("ST DP single plan, 11 x %d grid, 5 intervals: %.1f us per call" % (n_s, us))
    # SL path DP, default lattice, straight line, 3 static obstacles
    s_layers = np.array([[8.0, 16.0, 28.0]])
    sb = mb.SLBatch(s_layers, np.linspace(-1.45, 1.45, 5)[None], np.zeros((1, 2)),
                    np.array([[9.5, 16.0, 21.5]]), np.array([[0.2, -1.0, 1.2]]), np.array([3], np.int32), np.array([1.0]))
    out = mb.SLPlanOut(1, 3, 5, want_maps=False)
    prm = mb.sl_params(n_l=5)
    print("SL DP single plan, 3 x 5 lattice, 3 obstacles: %.1f us per call" % wall(lambda: mb.sl_plan(sb, prm, out)))
    # whole replan of one scene (curved line, 192 static + 64 moving obstacles)
    for B in (1, 16, 256):
        sc = bench.make_scenes(B)
        dev = {k: mb.to_dev(v, torch.int32 if v.dtype == np.int32 else torch.float64) for k, v in sc.items()}
        dev["line_vec"] = torch.zeros(B, dtype=torch.float64, device="cuda")
        params = pl.default_params()
        dims = pl.plan_dims(B, sc["knots_x"].shape[1], 192, 64, n_l=5, m_max=200, params=params)
        arena = pl.PlanArena(dims)
        us = wall(lambda: pl.plan_scenes(dev, dims, params, arena), reps=100)
        # the call only enqueues on the current stream, so it can be captured in a HIP graph
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pl.plan_scenes(dev, dims, params, arena)
        us_g = wall(g.replay, reps=100)
        print("whole replan, %d scene(s) per call, 256 obstacles each: %.1f us per call, %.1f us as a replayed "
              "HIP graph" % (B, us, us_g))


if __name__ == "__main__":
    main()
