"""Ad-hoc timing of the materialised SL path (edge tensor + DP kernel) on the bench's
lattice (development aid): mp_sl_edge_cost writes every edge cost to HBM (fp64,
8 B/edge), mp_sl_dp streams the tensor once.

    python3 tools/time_sl_split.py [B]
"""
import sys

import torch

sys.path.insert(0, '.')
from motion_planning_amd import batch as mb
from bench import N_L, N_S, make_scenarios


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    sc = make_scenarios(0, B)
    sb = mb.SLBatch(sc["s_layers"], sc["l_row"], sc["ego"], sc["ob_s"], sc["ob_l"], sc["n_ob"], sc["ob_dist"])
    prm = mb.sl_params(early_exit=False, n_l=N_L)
    edge, st = mb.sl_edge_cost(sb, prm)
    out = mb.sl_dp(edge, prm)
    fused = mb.sl_plan(sb, prm)
    torch.cuda.synchronize()
    assert torch.equal(out.path_idx, fused.path_idx) and torch.equal(out.status, fused.status)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_e = t_d = 0.0
    reps = 5
    for _ in range(reps):
        ev[0].record()
        edge, st = mb.sl_edge_cost(sb, prm)
        ev[1].record()
        out = mb.sl_dp(edge, prm, out=out)
        ev[2].record()
        torch.cuda.synchronize()
        t_e += ev[0].elapsed_time(ev[1]) / reps
        t_d += ev[1].elapsed_time(ev[2]) / reps
    n_edges = B * N_S * N_L * N_L
    gb = n_edges * 8 / 1e9
    print("B=%d: edge tensor %.1f GB; mp_sl_edge_cost %.3f ms (%.2f TB/s written, %.2e edges/s); "
          "mp_sl_dp %.3f ms (%.2f TB/s read)" % (B, gb, t_e, gb / t_e, n_edges / t_e * 1e3, t_d, gb / t_d))


if __name__ == "__main__":
    main()
