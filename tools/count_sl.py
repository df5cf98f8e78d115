"""Development aid: lattice edges the fused SL kernel prices per layer, per scenario
(counting instance, one scenario per launch)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from motion_planning_amd import batch as mb
from bench import N_L, N_S, make_scenarios

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sc = make_scenarios(0, n)
prm = mb.sl_params(early_exit=False, n_l=N_L)
rows, srows = [], []
for b in range(n):
    cnt, ser = [], []
    for L in (1, 2, 3, 4, 6, 64):
        sb = mb.SLBatch(sc["s_layers"][b:b+1], sc["l_row"][b:b+1], sc["ego"][b:b+1], sc["ob_s"][b:b+1],
                        sc["ob_l"][b:b+1], sc["n_ob"][b:b+1], sc["ob_dist"][b:b+1], n_layers=np.array([L], np.int32))
        _, c, cs = mb.sl_plan_counted(sb, prm)
        cnt.append(c)
        ser.append(cs)
    rows.append(cnt)
    srows.append(ser)
r = np.array(rows)
per = np.diff(r, axis=1)
print("layers<=1,2,3,4,6,64 cumulative mean:", r.mean(0))
print("layer 1 edges per scenario: mean %.0f median %.0f max %d; scenarios above 2000: %d of %d" % (
    per[:, 0].mean(), np.median(per[:, 0]), per[:, 0].max(), (per[:, 0] > 2000).sum(), n))
print("layer 1 per scenario:", per[:, 0].tolist())
print("layer 2 per scenario:", per[:, 1].tolist())
sr = np.array(srows)
sper = np.diff(sr, axis=1)
print("serial (busiest lane per wave, summed over 2 waves) cumulative mean:", sr.mean(0))
print("layer 1 serial per scenario:", sper[:, 0].tolist())
print("layer 2 serial per scenario:", sper[:, 1].tolist())
