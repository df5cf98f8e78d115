"""Ad-hoc timing of the fused SL kernel over batch sizes (development aid): the
time-vs-batch staircase shows how many workgroups a CU holds at once."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from motion_planning_amd import batch as mb
import bench

sc = bench.make_scenarios(0, 4096)
prm = mb.sl_params(early_exit=False, n_l=bench.N_L)
for B in [int(x) for x in (sys.argv[1:] or "256 1024 1536 1792 2048 2304 3584 4096".split())]:
    sb = mb.SLBatch(sc["s_layers"][:B], sc["l_row"][:B], sc["ego"][:B], sc["ob_s"][:B], sc["ob_l"][:B],
                    sc["n_ob"][:B], sc["ob_dist"][:B])
    out = mb.SLPlanOut(B, bench.N_S, bench.N_L, want_maps=False)
    for _ in range(3):
        mb.sl_plan(sb, prm, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        mb.sl_plan(sb, prm, out)
    e1.record(); torch.cuda.synchronize()
    print("B=%d: %.3f ms" % (B, e0.elapsed_time(e1) / 10))
