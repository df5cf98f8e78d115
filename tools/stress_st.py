"""One-off stress of the ST DP (development aid): random grids, interval counts, interval
layouts (nested, touching grid values, ends before starts, far ahead, behind) and start
speeds; the lean instances (no tables; per interval bound for large batches) against the
kernel that writes the tables, and a sample of every case against the CPU oracle's tables.

    python3 tools/stress_st.py [cases] [seed]
"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from motion_planning_amd import _lib, batch as mb
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lim = 25.0 * 5.0 / 18.0
bad = 0
for c in range(cases):
    B = int(rng.choice([64, 600, 2048, 2304, 4096]))
    n_t = int(rng.choice([3, 6, 11, 11, 11]))
    N = int(rng.choice([1, 2, 5, 29, 48, 49, 64, 97, 101, 128, 129, 200, 333]))
    r_max = int(rng.choice([0, 1, 2, 3, 5, 5, 6, 7, 8, 9, 12]))
    n_s = rng.integers(max(1, N // 2), N + 1, B).astype(np.int32)
    n_s[rng.integers(0, B)] = N
    R = max(r_max, 1)
    down = rng.uniform(-10, N + 20, (B, n_t, R))
    width = rng.uniform(-2.0, 8.0, (B, n_t, R))          # some intervals end before they start
    snap = rng.random((B, n_t, R)) < 0.3
    down = np.where(snap, np.round(down), down)           # interval starts on grid values
    up = down + np.where(snap, np.round(width), width)
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32) if r_max else np.zeros((B, n_t), np.int32)
    vel = rng.uniform(0.0, 12.0, B)
    full = mb.st_dp(n_s, down, up, n_reg, vel, lim, want_tables=True, n_s_max=N)
    # the first sweep's cells: the library's choice, none, or any number; on the instance for the
    # interval bound or on the kernel that holds every path
    prefix = int(rng.choice([-1, -1, 0, int(rng.integers(1, max(2, N)))]))
    generic = int(rng.choice([-1, -1, 1]))
    _lib.set_knob("MP_ST_PREFIX", prefix)
    _lib.set_knob("MP_ST_GENERIC", generic)
    lean = mb.st_dp(n_s, down, up, n_reg, vel, lim, want_tables=False, n_s_max=N)
    lean_o = mb.st_dp(n_s, down, up, n_reg, vel, lim, want_tables=False, n_s_max=N, ordered=True)
    _lib.set_knob("MP_ST_PREFIX", -1)
    _lib.set_knob("MP_ST_GENERIC", -1)
    torch.cuda.synchronize()
    ok = all(torch.equal(full[k], x[k]) for x in (lean, lean_o) for k in ("path_s_ind", "path_len"))
    cost, pre = full["cost"].cpu().numpy(), full["pre"].cpu().numpy()
    for b in rng.choice(B, 4, replace=False):
        n = int(n_s[b])
        ref = orc.st_find_speed(n_t, n, down[b], up[b], n_reg[b], vel[b], lim)
        ok = ok and np.array_equal(cost[b, :, :n], ref["cost"]) and np.array_equal(pre[b, :, :n], ref["pre"])
        plen = int(full["path_len"][b])
        ok = ok and np.array_equal(full["path_s_ind"][b, :plen].cpu().numpy(), ref["path_s_ind"])
    print("case %2d: B=%d n_t=%d N=%d r_max=%d prefix=%d generic=%d %s" %
          (c, B, n_t, N, r_max, prefix, generic, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
