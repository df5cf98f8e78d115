"""Experiment: SL plan and ST DP of successive batches on two streams (ST of batch k
waits for SL of batch k, SL of batch k+1 does not wait for ST of batch k) against
the same launches on one stream (development aid).

    python3 tools/time_overlap.py [B] [steps]
"""
import sys
import time

import torch

sys.path.insert(0, '.')
from motion_planning_amd import batch as mb
from bench import N_L, N_S, ST_N_S, SPEED_LIM, make_scenarios


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    sc = make_scenarios(0, B)
    sb = mb.SLBatch(sc["s_layers"], sc["l_row"], sc["ego"], sc["ob_s"], sc["ob_l"], sc["n_ob"], sc["ob_dist"])
    prm = mb.sl_params(early_exit=False, n_l=N_L)
    out = mb.SLPlanOut(B, N_S, N_L, want_maps=False)
    st_in = dict(n_s=mb.to_dev(sc["st_n_s"], torch.int32), s_down=mb.to_dev(sc["down"]), s_up=mb.to_dev(sc["up"]),
                 n_reg=mb.to_dev(sc["n_reg"], torch.int32), cur_vel=mb.to_dev(sc["cur_vel"]))

    def st():
        mb.st_dp(st_in["n_s"], st_in["s_down"], st_in["s_up"], st_in["n_reg"], st_in["cur_vel"], SPEED_LIM,
                 want_tables=False, n_s_max=ST_N_S)

    for _ in range(3):
        mb.sl_plan(sb, prm, out)
        st()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        mb.sl_plan(sb, prm, out)
        st()
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / steps
    sa, sb_ = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        with torch.cuda.stream(sa):
            mb.sl_plan(sb, prm, out)
            ev = torch.cuda.Event()
            ev.record(sa)
        with torch.cuda.stream(sb_):
            sb_.wait_event(ev)
            st()
    torch.cuda.synchronize()
    ovl = (time.perf_counter() - t0) / steps
    print("B=%d: one stream %.3f ms/step, two streams %.3f ms/step" % (B, seq * 1e3, ovl * 1e3))


if __name__ == "__main__":
    main()
