import sys, numpy as np, torch
sys.path.insert(0, '.')
from motion_planning_amd import batch as mb, pipeline as pl
import bench
B = 4096
sc = bench.make_scenes(B)
dev = {k: mb.to_dev(v, torch.int32 if v.dtype == np.int32 else torch.float64) for k, v in sc.items()}
dev["line_vec"] = torch.zeros(B, dtype=torch.float64, device="cuda")
params = pl.default_params()
dims = pl.plan_dims(B, sc["knots_x"].shape[1], 192, 64, n_l=5, m_max=200, params=params)
arena = pl.PlanArena(dims)
for _ in range(3):
    pl.plan_scenes(dev, dims, params, arena)
acc = None
for _ in range(5):
    _, ms = pl.plan_scenes_timed(dev, dims, params, arena)
    acc = ms if acc is None else [(n, a + b) for (n, a), (_, b) in zip(acc, ms)]
print(" | ".join("%s %.1f" % (n[:12], 1e3 * t / 5) for n, t in acc), "dims q_max", dims.q_max, "p_max", dims.p_max, "m_max", dims.m_max)
