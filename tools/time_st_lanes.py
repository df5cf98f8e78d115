"""The benchmark's speed DP (4096 scenarios, 11 x 97 grid, 5 intervals per layer) under the
launch knobs (development aid): lanes per cell x workgroup size.

    python3 tools/time_st_lanes.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from motion_planning_amd import _lib, batch as mb


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    sc = bench.make_scenarios(0, B)
    n_cells = int(os.environ.get("ST_AB_N", bench.ST_N_S))   # another grid width (what LDS holds changes with it)
    args = (mb.to_dev(np.minimum(sc["st_n_s"], n_cells), torch.int32), mb.to_dev(sc["down"]), mb.to_dev(sc["up"]),
            mb.to_dev(sc["n_reg"], torch.int32), mb.to_dev(sc["cur_vel"]), bench.SPEED_LIM)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(label):
        out = None
        for _ in range(3):
            out = mb.st_dp(*args, want_tables=False, n_s_max=n_cells, out=out)
        ts = []
        for _ in range(20):
            e0.record()
            out = mb.st_dp(*args, want_tables=False, n_s_max=n_cells, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%-34s %.4f ms (min %.4f)" % (label, float(np.median(ts)), min(ts)))
        return out

    ref = timed("library's choice")
    if os.environ.get("ST_AB_ONLY"):
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
