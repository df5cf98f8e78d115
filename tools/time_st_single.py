"""Latency of ONE speed plan (11 time layers, 5 moving intervals) at 29 / 61 / 101 cells under
the workgroup-size and lanes-per-cell knobs (development aid), as a replayed HIP graph.

    python3 tools/time_st_single.py
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from motion_planning_amd import _lib, batch as mb


def graph_us(fn, n=200):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))


def main():
    base = np.array([(12, 2), (20, 0), (6, 4), (25, -1), (15, 1)], dtype=np.float64)
    for s_end in (28.0, 60.0, 100.0):
        n_s = int(s_end) + 1
        s0 = base[:, 0] * (s_end / 28.0)
        down = (s0[None, :] + base[None, :, 1] * 0.5 * np.arange(11)[:, None])[None]
        args = (mb.to_dev(np.array([n_s]), torch.int32), mb.to_dev(down), mb.to_dev(down + 3.0),
                mb.to_dev(np.full((1, 11), 5), torch.int32), mb.to_dev(np.array([5.0])), bench.SPEED_LIM)
        o = mb.st_dp(*args, want_tables=False, n_s_max=n_s)
        fn = lambda: mb.st_dp(*args, want_tables=False, n_s_max=n_s, out=o)
        print("%3d cells: library's choice %.1f us" % (n_s, graph_us(fn)))
        for lanes in (4, 8):
            row = []
            for thr in (64, 128, 256, 512, 1024):
                _lib.set_knob("MP_ST_THREADS", thr)
                _lib.set_knob("MP_ST_LANES", lanes)
                row.append("%d: %.1f" % (thr, graph_us(fn)))
            print("    lanes per cell %d, threads -> us   %s" % (lanes, "   ".join(row)))
        _lib.set_knob("MP_ST_THREADS", -1)
        _lib.set_knob("MP_ST_LANES", -1)


if __name__ == "__main__":
    main()
