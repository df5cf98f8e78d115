"""Ad-hoc timing of the SL kernel on the reference's default lattice (5 lanes, layers
every 8 m) over obstacle counts (development aid).

    python3 tools/time_sl_narrow.py [B] [n_l]
"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from motion_planning_amd import batch as mb

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_l = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n_s = 12
rng = np.random.default_rng(3)
s_layers = np.broadcast_to(8.0 * (np.arange(n_s) + 1.0), (B, n_s)).copy()
l_row = np.broadcast_to(np.linspace(-1.45, 1.45, n_l), (B, n_l)).copy()
ego = np.stack([np.zeros(B), rng.uniform(-0.3, 0.3, B)], 1)
for n_ob in (0, 8, 64, 200, 900):
    ob_s = rng.uniform(2, 96, (B, max(n_ob, 1)))
    ob_l = rng.uniform(-4.3, 4.3, (B, max(n_ob, 1)))
    sb = mb.SLBatch(s_layers, l_row, ego, ob_s, ob_l, np.full(B, n_ob, np.int32), np.full(B, 0.8))
    prm = mb.sl_params(early_exit=False, n_l=n_l)
    out = mb.SLPlanOut(B, n_s, n_l, want_maps=False)
    for _ in range(3):
        mb.sl_plan(sb, prm, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        mb.sl_plan(sb, prm, out)
    e1.record(); torch.cuda.synchronize()
    print("B=%d n_l=%d n_ob=%d: %.3f ms, found %.2f" % (B, n_l, n_ob, e0.elapsed_time(e1) / 10,
                                                         float((out.status == 1).float().mean())))
