"""Ad-hoc timing of mp_refline_tables on the pipeline workload's reference lines
(development aid).  python3 tools/time_refline.py [B]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from motion_planning_amd import batch as mb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = bench.make_scenes(B)
kx, ky, nk = mb.to_dev(sc["knots_x"]), mb.to_dev(sc["knots_y"]), mb.to_dev(sc["n_knots"], torch.int32)
vec, ego = torch.zeros(B, dtype=torch.float64, device="cuda"), torch.zeros((B, 2), dtype=torch.float64, device="cuda")
for _ in range(3):
    t = mb.refline_tables(kx, ky, nk, vec, ego, m_max=200)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    t = mb.refline_tables(kx, ky, nk, vec, ego, m_max=200)
e1.record(); torch.cuda.synchronize()
print("refline_tables B=%d: %.3f ms (includes the table allocation)" % (B, e0.elapsed_time(e1) / 10))
