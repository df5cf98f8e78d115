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


from bench import make_scenes


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
