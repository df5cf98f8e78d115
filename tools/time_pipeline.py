"""Ad-hoc timing of the full planning pipeline (development aid): B robot-frame
scenes on curved reference lines with 192 static and 64 moving obstacles each,
wall time per call and, under rocprofv3 --kernel-trace --stats, time per kernel.

    python3 tools/time_pipeline.py [B] [n_l] [repeats]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from motion_planning_amd import pipeline as pl


def make_scenes(B, n_st=192, n_dy=64, seed=5):
    """Curved line y = c x^2 with 16..22 knots every 4.5 m, ego speed (U(3,8), U(-.5,.5)),
    obstacles spread over the line, most of them off the road so that a share of scenes
    keeps a free path."""
    rng = np.random.default_rng(seed)
    nk_max = 22
    kx, ky = np.zeros((B, nk_max)), np.zeros((B, nk_max))
    nk = rng.integers(16, 23, B).astype(np.int32)
    c = rng.uniform(-0.006, 0.006, B)
    for b in range(B):
        xs = -2.0 + 4.5 * np.arange(nk[b])
        kx[b, :nk[b]], ky[b, :nk[b]] = xs, c[b] * xs * xs
    x_end = -2.0 + 4.5 * (nk - 1)
    ego_vel = np.stack([rng.uniform(3, 8, B), rng.uniform(-0.5, 0.5, B)], 1)
    st_x = rng.uniform(0, 1, (B, n_st)) * x_end[:, None]
    off = rng.uniform(2.0, 14.0, (B, n_st)) * rng.choice([-1.0, 1.0], (B, n_st))
    st_pos = np.stack([st_x, c[:, None] * st_x * st_x + off], 2)
    dy_x = rng.uniform(0, 1, (B, n_dy)) * x_end[:, None]
    dy0 = np.stack([dy_x, c[:, None] * dy_x * dy_x + rng.uniform(2.5, 12.0, (B, n_dy)) *
                    rng.choice([-1.0, 1.0], (B, n_dy))], 2)
    vel = np.stack([rng.uniform(-3, 3, (B, n_dy)), rng.uniform(-0.3, 0.3, (B, n_dy))], 2)
    frames = [dy0]
    for _ in range(10):
        frames.append(frames[-1] + vel * 0.5)
    dy_pos = np.stack(frames, 2)
    return dict(knots_x=kx, knots_y=ky, n_knots=nk, ego_vel=ego_vel, st_pos=st_pos,
                st_rad=np.full((B, n_st), 0.8), n_st=np.full(B, n_st, np.int32), dy_pos=dy_pos,
                dy_rad=np.full((B, n_dy), 0.8), n_dy=np.full(B, n_dy, np.int32))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n_l = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    sc = make_scenes(B)
    dev = {k: torch.as_tensor(v).cuda() for k, v in sc.items()}
    r = pl.plan_batch(n_l=n_l, m_max=232, **dev)
    torch.cuda.synchronize()
    st = r["status"].cpu().numpy()
    print("B=%d n_l=%d: found %.3f, not found %.3f, other %.3f" %
          (B, n_l, (st == 1).mean(), (st == 0).mean(), ((st != 0) & (st != 1)).mean()))
    t0 = time.perf_counter()
    for _ in range(reps):
        r = pl.plan_batch(n_l=n_l, m_max=232, arena=r["arena"], **dev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("plan_batch: %.3f ms per call, %.0f plans/s" % (dt * 1e3, B / dt))


if __name__ == "__main__":
    main()
