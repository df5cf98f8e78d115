"""The speed DP of the whole-replan workload on its own (development aid): grid widths and
interval counts of the benchmark's scenes, and the kernel's time under the launch knobs.

    python3 tools/time_pipe_st.py [B]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from motion_planning_amd import _lib, batch as mb, pipeline as pl


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    sc = bench.make_scenes(B)
    dev = {k: mb.to_dev(v, torch.int32 if v.dtype == np.int32 else torch.float64) for k, v in sc.items()}
    dev["line_vec"] = torch.zeros(B, dtype=torch.float64, device="cuda")
    params = pl.default_params()
    dims = pl.plan_dims(B, sc["knots_x"].shape[1], 192, 64, n_l=5, m_max=200, params=params)
    arena = pl.plan_scenes(dev, dims, params)
    a = arena.arrays
    torch.cuda.synchronize()
    n_s = a["st_n_s"].cpu().numpy()
    n_reg = a["st_n_reg"].cpu().numpy()
    live = n_s > 0
    print("scenes with a speed search: %d of %d; cells: min %d median %d max %d" %
          (live.sum(), B, n_s[live].min(), np.median(n_s[live]), n_s[live].max()))
    print("cells histogram (bins of 10):", np.bincount(n_s[live] // 10))
    print("intervals per layer: mean %.2f, max %d; layers with none %.2f, with > 5: %.4f, > 8: %.4f" %
          (n_reg[live].mean(), n_reg[live].max(), (n_reg[live] == 0).mean(), (n_reg[live] > 5).mean(),
           (n_reg[live] > 8).mean()))
    args = (a["st_n_s"], a["st_down"], a["st_up"], a["st_n_reg"], a["robot_vel"], params.speed_lim)
    n_max = int(os.environ.get("PIPE_ST_NMAX", dims.st_n_s_max))   # a tighter bound on the grid width: less LDS
    print("n_s_max", n_max)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(label):
        out = None
        for _ in range(3):
            out = mb.st_dp(*args, want_tables=False, n_s_max=n_max, out=out, ordered=True)
        ts = []
        for _ in range(20):
            e0.record()
            out = mb.st_dp(*args, want_tables=False, n_s_max=n_max, out=out, ordered=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%-34s %.4f ms (min %.4f)" % (label, float(np.median(ts)), min(ts)))
        return out

    ref = timed("library's choice")
    if os.environ.get("ST_AB_ONLY"):
        _lib.set_knob("MP_ST_PREFIX", 0)
        o = timed("one sweep")
        _lib.set_knob("MP_ST_PREFIX", -1)
        assert torch.equal(o["path_s_ind"], ref["path_s_ind"]) and torch.equal(o["path_len"], ref["path_len"])
        return
    for thr in (128, 192, 256):
        for lanes in (4, 8):
            _lib.set_knob("MP_ST_THREADS", thr)
            _lib.set_knob("MP_ST_LANES", lanes)
            o = timed("threads %d, lanes per cell %d" % (thr, lanes))
            assert torch.equal(o["path_s_ind"], ref["path_s_ind"]) and torch.equal(o["path_len"], ref["path_len"])
    _lib.set_knob("MP_ST_THREADS", -1)
    _lib.set_knob("MP_ST_LANES", -1)


if __name__ == "__main__":
    main()
