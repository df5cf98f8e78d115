"""Ad-hoc timing of the ST DP kernel under different shapes (development aid)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from motion_planning_amd import batch as mb, _lib
import os
if os.environ.get('MP_LIB_PATH'):
    _lib.LIB_PATH = os.environ['MP_LIB_PATH']   # A/B timing against another build
import bench

def run(B, n_s, R, label):
    sc = bench.make_scenarios(0, B)
    n_reg = np.full((B, 11), R, np.int32)
    args = [mb.to_dev(np.full(B, n_s, np.int32), torch.int32), mb.to_dev(sc["down"]), mb.to_dev(sc["up"]),
            mb.to_dev(n_reg, torch.int32), mb.to_dev(sc["cur_vel"])]
    for _ in range(3):
        mb.st_dp(*args, bench.SPEED_LIM, want_tables=False, n_s_max=n_s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        mb.st_dp(*args, bench.SPEED_LIM, want_tables=False, n_s_max=n_s)
    e1.record(); torch.cuda.synchronize()
    print("%-30s B=%d n_s=%d R=%d: %.3f ms" % (label, B, n_s, R, e0.elapsed_time(e1) / 10))

run(4096, 97, 5, "bench shape")
run(4096, 97, 0, "no regions")
run(4096, 29, 5, "small grid")
run(4096, 29, 0, "small grid no regions")
run(1024, 97, 5, "quarter batch")
run(256, 97, 5, "256 scenarios")
