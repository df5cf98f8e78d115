"""ST speed DP and ST blocked intervals on the GPU (through the C ABI) against
the reference's golden outputs (bit-exact) and the CPU oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

LIM = 25.0 * 5.0 / 18.0


def same(a, b, what=""):
    """Bit-exact comparison with a short failure message."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = np.argwhere(a != b)
    assert len(bad) == 0, "%s: %d of %d differ, first at %s: %r vs %r" % (
        what, len(bad), a.size, bad[0].tolist(), a[tuple(bad[0])], b[tuple(bad[0])])


def st_cases():
    return [str(c) for c in load_golden("st.npz")["cases"]]


def test_st_golden_batch_bit_exact():
    """All golden ST cases in ONE ragged batch (different n_s and region counts)."""
    import torch
    from motion_planning_amd import batch as mb
    g = load_golden("st.npz")
    cases = st_cases()
    B = len(cases)
    n_s = np.array([int(g[c + "_n_s"]) for c in cases], dtype=np.int32)
    r_max = max(g[c + "_down"].shape[1] for c in cases)
    down = np.zeros((B, 11, r_max))
    up = np.zeros((B, 11, r_max))
    n_reg = np.zeros((B, 11), dtype=np.int32)
    cur = np.zeros(B)
    for b, c in enumerate(cases):
        r = g[c + "_down"].shape[1]
        down[b, :, :r] = g[c + "_down"]
        up[b, :, :r] = g[c + "_up"]
        n_reg[b] = g[c + "_n_reg"]
        cur[b] = float(g[c + "_cur_vel"])
    out = mb.st_dp(n_s, down, up, n_reg, cur, LIM)
    torch.cuda.synchronize()
    for b, c in enumerate(cases):
        n = n_s[b]
        same(out["cost"][b, :, :n].cpu().numpy(), g[c + "_cost"], c + " cost")
        same(out["pre"][b, :, :n].cpu().numpy(), g[c + "_pre"], c + " pre")
        plen = int(out["path_len"][b])
        assert np.array_equal(out["path_s_ind"][b, :plen].cpu().numpy(), g[c + "_path_s_ind"]), c
        assert (out["path_s_ind"][b, plen:].cpu().numpy() == -1).all()
        # padding cells keep the initial values
        assert (out["cost"][b, :, n:].cpu().numpy() == 0).all() and (out["pre"][b, :, n:].cpu().numpy() == -1).all()


@pytest.mark.parametrize("n_s_hi,B", [(29, 64), (101, 48), (230, 6)])
def test_st_random_vs_oracle(oracle, n_s_hi, B):
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(n_s_hi)
    n_t, r_max = 11, 6
    n_s = rng.integers(max(3, n_s_hi // 3), n_s_hi + 1, B).astype(np.int32)
    n_s[0] = n_s_hi
    n_s[1] = 2 if B > 1 else n_s[1]
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    down = rng.uniform(0, n_s_hi, (B, n_t, r_max))
    up = down + rng.uniform(0, 8, (B, n_t, r_max))
    cur = rng.uniform(0, 10, B)
    out = mb.st_dp(n_s, down, up, n_reg, cur, LIM)
    out_nt = mb.st_dp(n_s, down, up, n_reg, cur, LIM, want_tables=False)
    torch.cuda.synchronize()
    for b in range(B):
        ref = oracle.st_find_speed(n_t, int(n_s[b]), down[b], up[b], n_reg[b], float(cur[b]), LIM)
        n = n_s[b]
        same(out["cost"][b, :, :n].cpu().numpy(), ref["cost"], "cost[%d]" % b)
        same(out["pre"][b, :, :n].cpu().numpy(), ref["pre"], "pre[%d]" % b)
        plen = int(out["path_len"][b])
        assert np.array_equal(out["path_s_ind"][b, :plen].cpu().numpy(), ref["path_s_ind"])
    assert torch.equal(out["path_s_ind"], out_nt["path_s_ind"]) and torch.equal(out["path_len"], out_nt["path_len"])


@pytest.mark.parametrize("n_s_hi,r_max", [(97, 5), (40, 5), (97, 0), (70, 1), (97, 3), (60, 7), (97, 8), (128, 4),
                                            (129, 4), (200, 2)])
def test_st_large_batch_instances_match_the_general_kernel(oracle, n_s_hi, r_max, knob):
    """Batches of 2048 or more scenarios with at most 8 interval slots per layer run on a
    kernel instance built for that bound (layers with fewer intervals are padded with
    intervals that add exact zeros) and for the grid's width class (4 lanes per cell up to 128
    cells, 8 beyond); the kernel that holds
    every path (knob MP_ST_GENERIC=1) must give the same paths, and the tables it writes
    must match the oracle on a sample."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(1000 * n_s_hi + r_max)
    B, n_t = 2048, 11
    n_s = rng.integers(max(3, n_s_hi // 2), n_s_hi + 1, B).astype(np.int32)
    n_s[0] = n_s_hi
    slots = max(r_max, 1)
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    n_reg[: B // 2] = r_max
    down = rng.uniform(0, n_s_hi, (B, n_t, slots))
    up = down + rng.uniform(0, 8, (B, n_t, slots))
    cur = rng.uniform(0, 10, B)
    fast = mb.st_dp(n_s, down, up, n_reg, cur, LIM, want_tables=False, n_s_max=n_s_hi)
    knob("MP_ST_GENERIC", 1)
    gen = mb.st_dp(n_s, down, up, n_reg, cur, LIM, want_tables=False, n_s_max=n_s_hi)
    tab = mb.st_dp(n_s, down, up, n_reg, cur, LIM, want_tables=True, n_s_max=n_s_hi)
    torch.cuda.synchronize()
    for k in ("path_s_ind", "path_len"):
        assert torch.equal(fast[k], gen[k]), k
        assert torch.equal(fast[k], tab[k]), k
    for b in list(range(6)) + [B // 2, B - 1]:
        ref = oracle.st_find_speed(n_t, int(n_s[b]), down[b].reshape(n_t, -1), up[b].reshape(n_t, -1), n_reg[b],
                                   float(cur[b]), LIM)
        plen = int(fast["path_len"][b])
        assert np.array_equal(fast["path_s_ind"][b, :plen].cpu().numpy(), ref["path_s_ind"]), b
        same(tab["cost"][b, :, :n_s[b]].cpu().numpy(), ref["cost"], "cost[%d]" % b)


@pytest.mark.parametrize("B", [256, 2048])
def test_st_grids_of_one_to_three_cells(oracle, B):
    """The narrowest grids (a path shorter than 3 m), with intervals far beyond them: every
    interval slot of their cells must be classified (a stale slot would add whatever LDS held)."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(B + 1)
    n_t, N, r_max = 11, 90, 5
    n_s = rng.integers(1, 4, B).astype(np.int32)
    n_s[-1] = N
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    down = rng.uniform(0, N, (B, n_t, r_max))
    up = down + rng.uniform(0, 8, (B, n_t, r_max))
    cur = rng.uniform(0, 10, B)
    for rep in range(3):
        lean = mb.st_dp(n_s, down, up, n_reg, cur, LIM, want_tables=False, n_s_max=N)
        tab = mb.st_dp(n_s, down, up, n_reg, cur, LIM, want_tables=True, n_s_max=N)
        torch.cuda.synchronize()
        assert torch.equal(lean["path_s_ind"], tab["path_s_ind"]) and torch.equal(lean["path_len"], tab["path_len"])
        cost = tab["cost"].cpu().numpy()
        for b in range(0, B, max(1, B // 128)):
            ref = oracle.st_find_speed(n_t, int(n_s[b]), down[b], up[b], n_reg[b], float(cur[b]), LIM)
            same(cost[b, :, :n_s[b]], ref["cost"], "cost[%d]" % b)
            plen = int(lean["path_len"][b])
            assert np.array_equal(lean["path_s_ind"][b, :plen].cpu().numpy(), ref["path_s_ind"]), b


@pytest.mark.parametrize("B", [1600, 2500])
def test_st_widest_first_order_changes_nothing(B):
    """mp_st_dp_ordered launches the workgroups widest grid first; paths and tables must be
    those of mp_st_dp bit for bit (scenarios without a grid included)."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(B)
    n_t, N, r_max = 11, 90, 4
    n_s = rng.integers(0, N + 1, B).astype(np.int32)
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    down = rng.uniform(0, N, (B, n_t, r_max))
    up = down + rng.uniform(0, 8, (B, n_t, r_max))
    cur = rng.uniform(0, 10, B)
    tables = B < 2048
    plain = mb.st_dp(n_s, down, up, n_reg, cur, LIM, want_tables=tables, n_s_max=N)
    ordered = mb.st_dp(n_s, down, up, n_reg, cur, LIM, want_tables=tables, n_s_max=N, ordered=True)
    torch.cuda.synchronize()
    order = ordered["order_ws"][B:].cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(B))
    key = ordered["order_ws"][:B].cpu().numpy()
    assert (np.diff(key[order]) <= 0).all() and key.max() > key.min()     # heaviest bin first
    assert n_s[order[:20]].mean() > n_s[order[-20:]].mean()
    for k in ("path_s_ind", "path_len"):
        assert torch.equal(plain[k], ordered[k]), k
    if tables:      # (a scenario without a grid leaves its table rows unwritten)
        live = torch.as_tensor(n_s > 0, device=plain["cost"].device)
        for k in ("cost", "pre"):
            assert torch.equal(plain[k][live], ordered[k][live]), k


@pytest.mark.parametrize("n_s_hi,B,r_max", [(150, 10, 12), (600, 3, 6), (1024, 2, 4)])
def test_st_many_intervals_and_wide_grids(oracle, n_s_hi, B, r_max):
    """More than 8 intervals on a layer, and grids too wide for the per-cell
    interval tables, take the thread-per-cell walk; intervals whose end precedes
    their start switch the monotone bound off.  Same bits either way."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(n_s_hi + r_max)
    n_t = 11
    n_s = np.full(B, n_s_hi, np.int32)
    n_s[-1] = n_s_hi - 37
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    n_reg[0, 2], n_reg[0, 3], n_reg[0, 4] = r_max, min(8, r_max), min(7, r_max)
    down = rng.uniform(0, n_s_hi, (B, n_t, r_max))
    up = down + rng.uniform(0, 12, (B, n_t, r_max))
    up[0, 5, 0] = down[0, 5, 0] - 6.0      # inverted interval
    cur = rng.uniform(0, 10, B)
    out = mb.st_dp(n_s, down, up, n_reg, cur, LIM)
    torch.cuda.synchronize()
    for b in range(B):
        ref = oracle.st_find_speed(n_t, int(n_s[b]), down[b], up[b], n_reg[b], float(cur[b]), LIM)
        n = n_s[b]
        same(out["cost"][b, :, :n].cpu().numpy(), ref["cost"], "cost[%d]" % b)
        same(out["pre"][b, :, :n].cpu().numpy(), ref["pre"], "pre[%d]" % b)
        plen = int(out["path_len"][b])
        assert np.array_equal(out["path_s_ind"][b, :plen].cpu().numpy(), ref["path_s_ind"])


def test_st_exact_ties_pick_the_first_predecessor(oracle):
    """A grid blocked everywhere: every cost carries multiples of 1e8, the small
    terms are rounded away and several predecessors tie exactly; the first one
    in the reference's scan order must win."""
    import torch
    from motion_planning_amd import batch as mb
    B, n_t = 3, 11
    n_s = np.array([33, 64, 97], np.int32)
    n_reg = np.ones((B, n_t), np.int32)
    down = np.full((B, n_t, 1), -1.0)
    up = np.full((B, n_t, 1), 5000.0)
    cur = np.array([0.0, 5.0, 7.0])
    out = mb.st_dp(n_s, down, up, n_reg, cur, LIM)
    torch.cuda.synchronize()
    tied = 0
    for b in range(B):
        ref = oracle.st_find_speed(n_t, int(n_s[b]), down[b], up[b], n_reg[b], float(cur[b]), LIM)
        n = int(n_s[b])
        same(out["cost"][b, :, :n].cpu().numpy(), ref["cost"], "cost[%d]" % b)
        same(out["pre"][b, :, :n].cpu().numpy(), ref["pre"], "pre[%d]" % b)
        # count the cells of layer 2 whose minimum is reached by more than one predecessor
        # (speed_planner.py:120-160 restated with Python floats, ds = 1, dt = 0.5)
        c1 = ref["cost"][1]
        for j in range(n):
            cs = []
            for p in range(j + 1):
                v = 1.0 * (j - p) * 1.0 / 0.5
                spd = 100 * v * v * 0.5 if v > LIM else -100 * ((v - LIM) / LIM) * 0.5
                acc = 1.0 * (j - 2 * p) * 1.0 / 0.5 ** 2
                jk = acc / 0.5
                cs.append((((1e8 + spd) + acc * acc * 0.5) + jk * jk * 0.5) + c1[p])
            assert min(cs) == ref["cost"][2][j] and cs.index(min(cs)) == ref["pre"][2][j]
            tied += cs.count(min(cs)) > 1
    assert tied > 0     # the data really contains exact ties


@pytest.mark.parametrize("n_s_max", [40, 97])
def test_st_front_term_switches_on_exactly_at_grid_values(oracle, n_s_max):
    """Interval starts on whole and half metres: with ds = 1 and dt = 0.5 the look-ahead
    4.5 v is a whole number of metres too, so `s >= s_down - 4.5 v` is met with equality
    on many edges -- the distance from which a cell's intervals can take the front formula
    has to be exact there.  Scenes mix intervals far ahead (never reached), just ahead,
    behind, and none at all."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(40 + n_s_max)
    B, n_t, r_max = 24, 11, 5
    n_s = rng.integers(n_s_max // 2, n_s_max + 1, B).astype(np.int32)
    n_s[0] = n_s_max
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    n_reg[1] = 0
    down = rng.integers(0, 2 * n_s_max, (B, n_t, r_max)) * 0.5
    down[2] += 3.0 * n_s_max                      # far ahead of every cell
    down[3] = -rng.integers(1, 60, (n_t, r_max)) * 0.5   # behind every cell
    up = down + rng.integers(0, 12, (B, n_t, r_max)) * 0.5
    cur = rng.integers(0, 10, B).astype(np.float64)
    out = mb.st_dp(n_s, down, up, n_reg, cur, LIM)
    torch.cuda.synchronize()
    for b in range(B):
        ref = oracle.st_find_speed(n_t, int(n_s[b]), down[b], up[b], n_reg[b], float(cur[b]), LIM)
        n = n_s[b]
        same(out["cost"][b, :, :n].cpu().numpy(), ref["cost"], "cost[%d]" % b)
        same(out["pre"][b, :, :n].cpu().numpy(), ref["pre"], "pre[%d]" % b)
        plen = int(out["path_len"][b])
        assert np.array_equal(out["path_s_ind"][b, :plen].cpu().numpy(), ref["path_s_ind"])


def test_st_bad_scenarios_do_not_disturb_their_neighbours():
    """NaN / infinite interval bounds, a NaN start speed, interval counts outside the arrays
    and grid widths outside the allocation in some scenarios: the call returns and the
    other scenarios come out bit for bit as in a clean batch."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(77)
    B, n_t, r_max, N = 10, 11, 5, 60
    n_s = rng.integers(20, N + 1, B).astype(np.int32)
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    down = rng.uniform(0, N, (B, n_t, r_max))
    up = down + rng.uniform(0, 8, (B, n_t, r_max))
    cur = rng.uniform(0, 10, B)
    clean = mb.st_dp(n_s, down, up, n_reg, cur, LIM, n_s_max=N)
    n2, r2, d2, u2, c2 = n_s.copy(), n_reg.copy(), down.copy(), up.copy(), cur.copy()
    d2[1, 3, 0] = np.nan
    u2[2, 4, 1] = np.inf
    d2[3, :, :] = -np.inf
    c2[4] = np.nan
    r2[5, 2], r2[5, 6] = -4, 10 ** 6
    n2[6], n2[7] = 0, N + 50
    dirty = mb.st_dp(n2, d2, u2, r2, c2, LIM, n_s_max=N)
    torch.cuda.synchronize()
    good = [0, 8, 9]
    for k in ("cost", "pre", "path_s_ind", "path_len"):
        assert torch.equal(clean[k][good], dirty[k][good]), k
    plen = dirty["path_len"].cpu().numpy()
    assert plen[6] == 0 and plen[7] == 0          # no grid: no path
    assert (dirty["path_s_ind"][6].cpu().numpy() == -1).all()


def test_st_other_grid_steps(oracle):
    """ds/dt other than the defaults go through the same tables."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(5)
    B, n_t, n_s, r_max = 8, 9, 40, 3
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    down = rng.uniform(0, 30, (B, n_t, r_max))
    up = down + rng.uniform(0, 5, (B, n_t, r_max))
    cur = rng.uniform(0, 10, B)
    out = mb.st_dp(np.full(B, n_s, np.int32), down, up, n_reg, cur, 7.3, ds=0.7, dt=0.3)
    torch.cuda.synchronize()
    for b in range(B):
        ref = oracle.st_find_speed(n_t, n_s, down[b], up[b], n_reg[b], float(cur[b]), 7.3, ds=0.7, dt=0.3)
        same(out["cost"][b].cpu().numpy(), ref["cost"], "cost[%d]" % b)
        same(out["pre"][b].cpu().numpy(), ref["pre"], "pre[%d]" % b)


@pytest.mark.parametrize("lim,ds,dt", [(7.3, 0.7, 0.3), (0.4, 1.0, 0.5), (1e3, 1.0, 0.5), (25.0 * 5.0 / 18.0, 0.5, 0.5),
                                       (3.0, 2.0, 0.25)])
def test_st_large_batches_with_other_steps_and_limits(oracle, knob, lim, ds, dt):
    """A large batch (planned in two sweeps, the first over the cells a vehicle within the speed
    limit reaches) with other grid steps and speed limits: a limit below the first step (every
    edge but standing still is above it), one no cell can exceed (no cell is ever too far
    ahead), half-metre cells.  Against the kernel that writes the tables, the thread-per-cell
    relaxation and, on a sample, the oracle."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(int(lim * 100) + int(ds * 10))
    B, n_t, n_s_hi, r_max = 2560, 11, 90, 4
    n_s = rng.integers(20, n_s_hi + 1, B).astype(np.int32)
    n_s[:64] = n_s_hi
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    down = rng.uniform(-3.0, n_s_hi * ds + 5.0, (B, n_t, r_max))
    up = down + rng.uniform(0.0, 6.0, down.shape)
    cur = rng.uniform(0.0, 2.0 * min(lim, 10.0), B)
    args = (mb.to_dev(n_s, torch.int32), mb.to_dev(down), mb.to_dev(up), mb.to_dev(n_reg, torch.int32), mb.to_dev(cur))
    ref = mb.st_dp(*args, lim, ds=ds, dt=dt, want_tables=True, n_s_max=n_s_hi)
    fast = mb.st_dp(*args, lim, ds=ds, dt=dt, want_tables=False, n_s_max=n_s_hi)
    knob("MP_ST_PREFIX", 11)
    fast_k = mb.st_dp(*args, lim, ds=ds, dt=dt, want_tables=False, n_s_max=n_s_hi)
    knob("MP_ST_GENERIC", 1)
    gen_k = mb.st_dp(*args, lim, ds=ds, dt=dt, want_tables=False, n_s_max=n_s_hi)
    knob("MP_ST_PREFIX", -1)
    knob("MP_ST_GENERIC", 2)
    cell = mb.st_dp(*args, lim, ds=ds, dt=dt, want_tables=False, n_s_max=n_s_hi)
    knob("MP_ST_GENERIC", -1)
    torch.cuda.synchronize()
    for name, o in (("library's choice", fast), ("first sweep over 12 cells", fast_k),
                    ("every path, first sweep over 12 cells", gen_k), ("thread per cell", cell)):
        assert torch.equal(o["path_s_ind"], ref["path_s_ind"]) and torch.equal(o["path_len"], ref["path_len"]), name
    for b in (0, 1, 64, B - 1):
        o = oracle.st_find_speed(n_t, int(n_s[b]), down[b], up[b], n_reg[b], float(cur[b]), lim, ds=ds, dt=dt)
        plen = int(fast["path_len"][b])
        assert np.array_equal(fast["path_s_ind"][b, :plen].cpu().numpy(), o["path_s_ind"]), b
        same(ref["cost"][b, :, :n_s[b]].cpu().numpy(), o["cost"], "cost[%d]" % b)


@pytest.mark.parametrize("B,n_t,N,r_max", [(2048, 11, 1000, 3), (2048, 40, 300, 2), (600, 11, 1024, 8),
                                           (2304, 64, 120, 5), (4096, 3, 97, 5)])
def test_st_extreme_shapes_against_thread_per_cell(knob, B, n_t, N, r_max):
    """The widest grids, the most and the fewest time layers the library takes, in large batches
    (two sweeps, records and tables sized by the first one): the instance for the interval bound
    and the kernel that holds every path against the relaxation by one thread per cell."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(B + n_t + N)
    n_s = rng.integers(N // 2, N + 1, B).astype(np.int32)
    n_s[0] = N
    down = rng.uniform(-10.0, N + 20.0, (B, n_t, r_max))
    up = down + rng.uniform(0.0, 8.0, down.shape)
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    cur = rng.uniform(0.0, 12.0, B)
    args = (mb.to_dev(n_s, torch.int32), mb.to_dev(down), mb.to_dev(up), mb.to_dev(n_reg, torch.int32), mb.to_dev(cur))
    fast = mb.st_dp(*args, LIM, want_tables=False, n_s_max=N)
    knob("MP_ST_GENERIC", 1)
    gen = mb.st_dp(*args, LIM, want_tables=False, n_s_max=N)
    knob("MP_ST_GENERIC", 2)
    ref = mb.st_dp(*args, LIM, want_tables=False, n_s_max=N)
    knob("MP_ST_GENERIC", -1)
    torch.cuda.synchronize()
    assert int(ref["path_len"].min()) >= 2
    for name, o in (("library's choice", fast), ("every path", gen)):
        assert torch.equal(o["path_s_ind"], ref["path_s_ind"]) and torch.equal(o["path_len"], ref["path_len"]), name


def test_st_regions_pipeline_goldens():
    import torch
    from motion_planning_amd import batch as mb
    g = load_golden("pipe.npz")
    cases = [str(c) for c in g["cases"] if bool(g[str(c) + "_found"])]
    B = len(cases)
    n_t = 11
    p_max = max(len(g[c + "_path_sl"]) for c in cases)
    n_st_max = max(1, max(len(g[c + "_st_traj"]) for c in cases))
    n_dy_max = max(1, max(len(g[c + "_dy_traj"]) for c in cases))
    ps, pl = np.zeros((B, p_max)), np.zeros((B, p_max))
    npath = np.zeros(B, np.int32)
    st_s, st_l, st_r = (np.zeros((B, n_st_max)) for _ in range(3))
    n_st = np.zeros(B, np.int32)
    dy_s, dy_l = np.zeros((B, n_dy_max, n_t)), np.zeros((B, n_dy_max, n_t))
    dy_r = np.zeros((B, n_dy_max))
    n_dy = np.zeros(B, np.int32)
    for b, c in enumerate(cases):
        path = g[c + "_path_sl"]
        npath[b] = len(path)
        ps[b, :len(path)], pl[b, :len(path)] = path[:, 0], path[:, 1]
        st = g[c + "_st_traj"]
        n_st[b] = len(st)
        st_s[b, :len(st)], st_l[b, :len(st)], st_r[b, :len(st)] = st[:, 0], st[:, 1], g[c + "_st_rad"]
        dy = g[c + "_dy_traj"]
        n_dy[b] = len(dy)
        dy_s[b, :len(dy)], dy_l[b, :len(dy)], dy_r[b, :len(dy)] = dy[:, :, 0], dy[:, :, 1], g[c + "_dy_rad"]
    r_max = 8
    down, up, n_reg, status = mb.st_regions(n_t, ps, pl, npath, st_s, st_l, st_r, n_st, dy_s, dy_l, dy_r, n_dy, r_max)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 1).all()
    for b, c in enumerate(cases):
        want_n = g[c + "_st_n_reg"]
        assert np.array_equal(n_reg[b].cpu().numpy(), want_n), c
        for i in range(n_t):
            assert np.array_equal(down[b, i, :want_n[i]].cpu().numpy(), g[c + "_st_down"][i, :want_n[i]])
            assert np.array_equal(up[b, i, :want_n[i]].cpu().numpy(), g[c + "_st_up"][i, :want_n[i]])


@pytest.mark.parametrize("once", [True, False], ids=["scan_once", "scan_per_layer"])
@pytest.mark.parametrize("sizes", [(300, 40, 64), (3, 2, 4), (64, 0, 64), (0, 9, 8)], ids=str)
def test_st_regions_random_vs_oracle(oracle, sizes, once, knob):
    """Both kernels behind mp_st_regions: every (obstacle, layer) scanned once with the
    results kept on chip, and the plain scan per layer used when they do not fit."""
    import torch
    from motion_planning_amd import batch as mb
    if not once:
        knob("MP_ST_REGIONS_ONCE", 0)
    rng = np.random.default_rng(77)
    B, n_t, p_max = 12, 11, 150
    n_st_max, n_dy_max, r_max = sizes
    npath = rng.integers(1, p_max + 1, B).astype(np.int32)
    ps = np.cumsum(np.full((B, p_max), 0.5), axis=1) - 0.5
    pl = 0.8 * np.sin(ps / 7.0 + rng.uniform(0, 3, (B, 1)))
    n_st = rng.integers(0, n_st_max + 1, B).astype(np.int32)
    n_st[0], n_st[1] = 0, n_st_max
    n_dy = rng.integers(0, n_dy_max + 1, B).astype(np.int32)
    n_dy[0] = 0
    npath[2] = 1
    st_s, st_l = rng.uniform(-5, 80, (B, n_st_max)), rng.uniform(-6, 6, (B, n_st_max))
    st_r = rng.uniform(0.2, 1.5, (B, n_st_max))
    dy_s = rng.uniform(-5, 80, (B, n_dy_max, 1)) + rng.uniform(-3, 3, (B, n_dy_max, 1)) * np.arange(n_t) * 0.5
    dy_l = rng.uniform(-4, 4, (B, n_dy_max, 1)) + rng.uniform(-1, 1, (B, n_dy_max, 1)) * np.arange(n_t) * 0.5
    dy_r = rng.uniform(0.2, 1.5, (B, n_dy_max))
    down, up, n_reg, status = mb.st_regions(n_t, ps, pl, npath, st_s, st_l, st_r, n_st, dy_s, dy_l, dy_r, n_dy, r_max)
    torch.cuda.synchronize()
    for b in range(B):
        path = np.stack([ps[b, :npath[b]], pl[b, :npath[b]]], 1)
        st = [(np.array([st_s[b, o], st_l[b, o]]), st_r[b, o]) for o in range(n_st[b])]
        dy = [(np.stack([dy_s[b, o], dy_l[b, o]], 1), dy_r[b, o]) for o in range(n_dy[b])]
        rd, ru, rn = oracle.st_add_obstacle(n_t, path, st, dy)
        if rn.max() > r_max:
            assert int(status[b]) == -3
            continue
        assert int(status[b]) == 1
        assert np.array_equal(n_reg[b].cpu().numpy(), rn)
        for i in range(n_t):
            assert np.array_equal(down[b, i, :rn[i]].cpu().numpy(), rd[i, :rn[i]])
            assert np.array_equal(up[b, i, :rn[i]].cpu().numpy(), ru[i, :rn[i]])


@pytest.mark.parametrize("n_s_hi,r_max,B", [(97, 5, 8192), (101, 8, 4096), (60, 3, 8192), (128, 2, 4096),
                                             (200, 6, 1024), (33, 12, 4096)])
def test_st_many_random_scenarios_every_relaxation_agrees(knob, n_s_hi, r_max, B):
    """Thousands of random ST grids per shape (widths 2..n_s_hi, 0..r_max intervals per layer,
    some with their end before their start, intervals sitting exactly on grid values): the
    instance built for the interval bound, the kernel that holds every path (with and without
    the cost / predecessor tables written), both lane splits and the relaxation by one thread
    per cell give the same plans, and the same tables where they are written, bit for bit."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(500 + n_s_hi + r_max)
    n_t = 11
    n_s = rng.integers(2, n_s_hi + 1, B).astype(np.int32)
    n_s[: B // 4] = n_s_hi
    down = rng.uniform(-5.0, n_s_hi + 10.0, (B, n_t, max(r_max, 1)))
    snap = rng.random(down.shape) < 0.2
    down[snap] = np.round(down[snap])                       # on grid values
    up = down + rng.uniform(0.0, 6.0, down.shape)
    flip = rng.random(down.shape) < 0.03
    up[flip] = down[flip] - rng.uniform(0.0, 2.0, int(flip.sum()))   # an end before its start
    n_reg = rng.integers(0, r_max + 1, (B, n_t)).astype(np.int32)
    n_reg[rng.random(B) < 0.3] = 0
    cur = rng.uniform(0.0, 12.0, B)
    args = (mb.to_dev(n_s, torch.int32), mb.to_dev(down), mb.to_dev(up), mb.to_dev(n_reg, torch.int32), mb.to_dev(cur))
    ref = mb.st_dp(*args, LIM, want_tables=True, n_s_max=n_s_hi)
    outs = {"instance / library's choice": mb.st_dp(*args, LIM, want_tables=False, n_s_max=n_s_hi)}
    for lanes in (4, 8):
        knob("MP_ST_LANES", lanes)
        outs["%d lanes per cell" % lanes] = mb.st_dp(*args, LIM, want_tables=False, n_s_max=n_s_hi)
        knob("MP_ST_GENERIC", 1)
        outs["every path, %d lanes" % lanes] = mb.st_dp(*args, LIM, want_tables=False, n_s_max=n_s_hi)
        outs["every path, tables, %d lanes" % lanes] = mb.st_dp(*args, LIM, want_tables=True, n_s_max=n_s_hi)
        knob("MP_ST_GENERIC", -1)
    knob("MP_ST_LANES", -1)
    # the first sweep's cells: none (one sweep), a few (a loose bound on the plan's cost: most cells
    # of the second sweep are walked), most of the grid
    for k in (0, 3, n_s_hi // 3, n_s_hi - 10):
        knob("MP_ST_PREFIX", k)
        outs["first sweep over %d cells" % (k + 1)] = mb.st_dp(*args, LIM, want_tables=False, n_s_max=n_s_hi)
        knob("MP_ST_GENERIC", 1)
        outs["every path, first sweep over %d cells" % (k + 1)] = mb.st_dp(*args, LIM, want_tables=False, n_s_max=n_s_hi)
        knob("MP_ST_GENERIC", -1)
    knob("MP_ST_PREFIX", -1)
    knob("MP_ST_GENERIC", 2)
    outs["thread per cell"] = mb.st_dp(*args, LIM, want_tables=False, n_s_max=n_s_hi)
    outs["thread per cell, tables"] = mb.st_dp(*args, LIM, want_tables=True, n_s_max=n_s_hi)
    knob("MP_ST_GENERIC", -1)
    torch.cuda.synchronize()
    assert int(ref["path_len"].min()) >= 2
    for name, o in outs.items():
        bad = ((o["path_s_ind"] != ref["path_s_ind"]).any(1) | (o["path_len"] != ref["path_len"])).nonzero().flatten()
        assert bad.numel() == 0, "%s: plans of scenarios %s differ" % (name, bad[:8].cpu().numpy())
        if o.get("cost") is not None:
            assert torch.equal(o["cost"], ref["cost"]) and torch.equal(o["pre"], ref["pre"]), name
