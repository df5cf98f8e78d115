"""Development aid: plan the first N benchmark scenarios with the lean fused SL kernel, with the
counting instance and with every predecessor row in LDS; print the scenarios where they differ
and what the CPU oracle says about those.

    python3 tools/find_sl_diff.py [N]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from motion_planning_amd import _lib
from motion_planning_amd import batch as mb


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    sc = bench.make_scenarios(0, N)
    prm = mb.sl_params(early_exit=False, n_l=bench.N_L)
    sb = mb.SLBatch(sc["s_layers"], sc["l_row"], sc["ego"], sc["ob_s"], sc["ob_l"], sc["n_ob"], sc["ob_dist"])
    lean = mb.SLPlanOut(N, bench.N_S, bench.N_L, want_maps=False)
    mb.sl_plan(sb, prm, lean)
    torch.cuda.synchronize()
    cnt, _, _ = mb.sl_plan_counted(sb, prm)
    _lib.set_knob("MP_SL_ROWS", 255)
    full = mb.SLPlanOut(N, bench.N_S, bench.N_L, want_maps=False)
    mb.sl_plan(sb, prm, full)
    torch.cuda.synchronize()
    for name, o in (("counted", cnt), ("all rows", full)):
        d = ((o.path_idx != lean.path_idx).any(1) | (o.status != lean.status)).nonzero().flatten().cpu().numpy()
        print("lean vs %s: %d scenarios differ: %s" % (name, len(d), d[:16]))
    d = ((cnt.path_idx != lean.path_idx).any(1) | (cnt.status != lean.status)).nonzero().flatten().cpu().numpy()
    if not len(d):
        return
    from oracle import oracle as orc
    orc.build()
    orc.lib()
    for b in d[:4]:
        b = int(b)
        print("scenario", b, "status lean/counted/full", int(lean.status[b]), int(cnt.status[b]), int(full.status[b]),
              "cost", float(lean.best_cost[b]), float(cnt.best_cost[b]), float(full.best_cost[b]))
        print("  lean   ", lean.path_idx[b].cpu().numpy())
        print("  counted", cnt.path_idx[b].cpu().numpy())
        print("  full   ", full.path_idx[b].cpu().numpy())
        s_map = np.repeat(sc["s_layers"][b][:, None], bench.N_L, axis=1)
        l_map = np.broadcast_to(sc["l_row"][b], (bench.N_S, bench.N_L))
        ob = np.stack([sc["ob_s"][b], sc["ob_l"][b]], 1)
        r = orc.sl_find_path(s_map, l_map, int((bench.N_L - 1) / 2), sc["ego"][b], ob, sc["ob_dist"][b],
                             no_early_exit=True)
        print("  oracle ", r["path_idx"], bool(r["found"]), float(r["cost_map"][-1, r["path_idx"][-1]]))
        one = mb.SLBatch(sc["s_layers"][b:b + 1], sc["l_row"][b:b + 1], sc["ego"][b:b + 1], sc["ob_s"][b:b + 1],
                         sc["ob_l"][b:b + 1], sc["n_ob"][b:b + 1], sc["ob_dist"][b:b + 1])
        ocm = r["cost_map"]
        for cap in (-1, 64, 128, 200, 256, 400, 512, 1024, 2048):
            for rows in (-1, 255):
                _lib.set_knob("MP_SL_PAIR_CAP", cap)
                _lib.set_knob("MP_SL_ROWS", rows)
                o1 = mb.SLPlanOut(1, bench.N_S, bench.N_L, want_maps=False)
                mb.sl_plan(one, prm, o1)
                om = mb.SLPlanOut(1, bench.N_S, bench.N_L, want_maps=True)
                mb.sl_plan(one, prm, om)
                torch.cuda.synchronize()
                cm = om.cost_map[0].cpu().numpy()
                bad = np.argwhere(cm != ocm)
                print("  cap %5d rows %3d: plain status %d cost %.3f | maps status %d, cost map differs from the oracle's "
                      "at %d nodes, first %s" % (cap, rows, int(o1.status[0]), float(o1.best_cost[0]), int(om.status[0]),
                                                 len(bad), bad[0] if len(bad) else None))
        print("  pairs per layer (oracle view): n_ob", int(sc["n_ob"][b]), "radius", float(sc["ob_dist"][b]))
        print("  layer minima (oracle):", np.round(ocm.min(1)[:12], 1))

if __name__ == "__main__":
    main()
