"""Experiment: a step's batch cut into parts that run SL plan -> ST DP each, the parts
alternating between two streams, against the whole batch on one stream (development aid).

    python3 tools/time_split.py [B] [steps]
"""
import sys
import time

import torch

sys.path.insert(0, '.')
from motion_planning_amd import batch as mb
from bench import N_L, N_S, ST_N_S, SPEED_LIM, make_scenarios


def part(sc, lo, hi):
    sb = mb.SLBatch(sc["s_layers"][lo:hi], sc["l_row"][lo:hi], sc["ego"][lo:hi], sc["ob_s"][lo:hi], sc["ob_l"][lo:hi],
                    sc["n_ob"][lo:hi], sc["ob_dist"][lo:hi])
    out = mb.SLPlanOut(hi - lo, N_S, N_L, want_maps=False)
    st_in = [mb.to_dev(sc["st_n_s"][lo:hi], torch.int32), mb.to_dev(sc["down"][lo:hi]), mb.to_dev(sc["up"][lo:hi]),
             mb.to_dev(sc["n_reg"][lo:hi], torch.int32), mb.to_dev(sc["cur_vel"][lo:hi])]
    return sb, out, st_in


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    sc = make_scenarios(0, B)
    prm = mb.sl_params(early_exit=False, n_l=N_L)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    join = len(sys.argv) > 3 and sys.argv[3] == "join"   # the streams meet after every step
    for n_parts in (1, 2, 3):
        size = B // n_parts
        parts = [part(sc, k * size, (k + 1) * size if k + 1 < n_parts else B) for k in range(n_parts)]
        for n_streams in ((1,) if n_parts == 1 else (n_parts,)):
            while len(streams) < n_streams:
                streams.append(torch.cuda.Stream())
            def step():
                evs = []
                for k, (sb, out, st_in) in enumerate(parts):
                    with torch.cuda.stream(streams[k % n_streams]):
                        mb.sl_plan(sb, prm, out)
                        mb.st_dp(*st_in, SPEED_LIM, want_tables=False, n_s_max=ST_N_S)
                        if join:
                            e = torch.cuda.Event(); e.record(); evs.append(e)
                if join:
                    for st_ in streams[:n_streams]:
                        for e in evs:
                            st_.wait_event(e)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            print("B=%d in %d part(s) on %d stream(s): %.3f ms/step" %
                  (B, n_parts, n_streams, (time.perf_counter() - t0) / steps * 1e3), flush=True)


if __name__ == "__main__":
    main()
