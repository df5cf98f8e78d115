"""Ad-hoc timing of the fused SL plan kernel on the bench's lattice workload
(development aid).  MP_LIB_PATH=<other build> times that library instead.

    python3 tools/time_sl.py [B] [repeats]
"""
import os
import sys

import torch

sys.path.insert(0, '.')
from motion_planning_amd import _lib
if os.environ.get('MP_LIB_PATH'):
    _lib.LIB_PATH = os.environ['MP_LIB_PATH']   # A/B timing against another build
from motion_planning_amd import batch as mb
from bench import N_L, N_S, make_scenarios


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    sc = make_scenarios(0, B)
    if os.environ.get('SL_NO_OB'):   # the same lattice without obstacles: the DP's fixed cost per layer
        sc["n_ob"] = sc["n_ob"] * 0
    if os.environ.get('SL_OB_SCALE'):   # shrink the collision radius: fewer pairs, fewer blocked edges
        sc["ob_dist"] = sc["ob_dist"] * float(os.environ['SL_OB_SCALE'])
    n_layers = None
    if os.environ.get('SL_LAYERS'):   # only the first k layers of every scenario: where the time goes
        import numpy as np
        n_layers = np.full(B, int(os.environ['SL_LAYERS']), np.int32)
    sb = mb.SLBatch(sc["s_layers"], sc["l_row"], sc["ego"], sc["ob_s"], sc["ob_l"], sc["n_ob"], sc["ob_dist"],
                    n_layers=n_layers)
    prm = mb.sl_params(early_exit=False, n_l=N_L)
    out = mb.SLPlanOut(B, N_S, N_L, want_maps=False)
    for _ in range(5):
        mb.sl_plan(sb, prm, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    tot = 0.0
    for _ in range(reps):
        e0.record()
        mb.sl_plan(sb, prm, out)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1)
        best = min(best, t)
        tot += t
    print("sl_plan B=%d: mean %.4f ms, best %.4f ms (%s)" % (B, tot / reps, best, _lib.LIB_PATH))


if __name__ == "__main__":
    main()
