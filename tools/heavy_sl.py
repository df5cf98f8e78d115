"""Which scenarios of the lattice benchmark are the long ones (development aid): every one of
the first N scenarios planned alone (one workgroup on an idle GPU), with the first k layers
only, and what the counting instance says about the longest.

    python3 tools/heavy_sl.py [N] [layers]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from motion_planning_amd import batch as mb


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    sc = bench.make_scenarios(0, N)
    prm = mb.sl_params(early_exit=False, n_l=bench.N_L)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    us = np.zeros(N)
    for b in range(N):
        sb = mb.SLBatch(sc["s_layers"][b:b + 1], sc["l_row"][b:b + 1], sc["ego"][b:b + 1], sc["ob_s"][b:b + 1],
                        sc["ob_l"][b:b + 1], sc["n_ob"][b:b + 1], sc["ob_dist"][b:b + 1],
                        n_layers=np.full(1, L, np.int32))
        out = mb.SLPlanOut(1, bench.N_S, bench.N_L, want_maps=False)
        mb.sl_plan(sb, prm, out)
        best = 1e9
        for _ in range(3):
            e0.record()
            mb.sl_plan(sb, prm, out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        us[b] = best
    order = np.argsort(-us)
    print("one scenario per call, %d layers: median %.1f us, p90 %.1f, p99 %.1f, max %.1f" %
          (L, np.median(us), np.percentile(us, 90), np.percentile(us, 99), us.max()))
    for b in order[:8]:
        sb = mb.SLBatch(sc["s_layers"][b:b + 1], sc["l_row"][b:b + 1], sc["ego"][b:b + 1], sc["ob_s"][b:b + 1],
                        sc["ob_l"][b:b + 1], sc["n_ob"][b:b + 1], sc["ob_dist"][b:b + 1],
                        n_layers=np.full(1, L, np.int32))
        o, n_eval, n_serial = mb.sl_plan_counted(sb, prm)
        near = np.sort(sc["ob_s"][b])[:4]
        print("scenario %4d: %.1f us; edges priced %d, one after the other %d, walk turns %d, collision tests %d; "
              "radius %.2f, nearest obstacles at s = %s"
              % (b, us[b], n_eval, n_serial, o.walk_turns, o.walk_tests, sc["ob_dist"][b], np.round(near, 2)))


if __name__ == "__main__":
    main()
