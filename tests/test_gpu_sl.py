"""SL path DP on the GPU (through the C ABI) against the reference's golden
outputs and against the CPU oracle on seeded random lattices."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import load_golden
from parity import assert_costs, assert_sl_parity, pad_obstacles

pytestmark = pytest.mark.gpu


def gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=True, split=False, shared_row=False,
           n_layers=None, per_node=False):
    """Runs B scenarios; s_map/l_map [B, n_s, n_l]. Returns list of dicts (numpy).
    per_node: lattices of up to 16 lanes normally take the lane-per-edge kernel; this
    sends them through the thread-per-node kernel instead."""
    import os
    import torch
    from motion_planning_amd import batch as mb
    s_map, l_map = np.asarray(s_map), np.asarray(l_map)
    B, n_s, n_l = s_map.shape
    ob_s, ob_l, n_ob = pad_obstacles(obs)
    sb = mb.SLBatch(s_map[:, :, 0], l_map[:, 0, :] if shared_row else l_map, ego, ob_s, ob_l, n_ob, ob_dist,
                    n_layers=n_layers)
    prm = mb.sl_params(early_exit=early_exit, mid_l=mid_l)
    if split:
        edge, st = mb.sl_edge_cost(sb, prm)
        assert (st.cpu().numpy() == 1).all()
        out = mb.sl_dp(edge, prm, n_layers=n_layers)
    else:
        from motion_planning_amd import _lib
        if per_node:
            _lib.set_knob("MP_SL_NARROW", 0)
        try:
            out = mb.sl_plan(sb, prm)
        finally:
            _lib.set_knob("MP_SL_NARROW", -1)
    torch.cuda.synchronize()
    res = []
    for b in range(B):
        res.append(dict(found=int(out.status[b]) == 1, status=int(out.status[b]),
                        cost_map=out.cost_map[b].cpu().numpy(), index_map=out.index_map[b].cpu().numpy(),
                        path_idx=out.path_idx[b].cpu().numpy(), layers_done=int(out.layers_done[b]),
                        best_cost=float(out.best_cost[b])))
    if split:
        return res, edge.cpu().numpy()
    return res


def golden_case(g, case):
    p = case + "_"
    ref = dict(found=bool(g[p + "found"]), cost_map=g[p + "cost_map"], index_map=g[p + "index_map"],
               path_idx=g[p + "path_idx"])
    # rows the reference never relaxed stay at their initial values
    done = int(np.sum(np.any(ref["index_map"] != -2, axis=1)))
    ref["layers_done"] = done
    inp = dict(s_map=g[p + "s_map"][None], l_map=g[p + "l_map"][None], mid_l=int(g[p + "mid_l"]),
               ego=g[p + "ego"][None], obs=[g[p + "ob"]], ob_dist=np.array([float(g[p + "ob_dist"])]))
    edge = g[p + "edge_cost"] if (p + "edge_cost") in g.files else None
    return inp, ref, edge


def sl_cases():
    return [str(c) for c in load_golden("sl.npz")["cases"]]


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("case", sl_cases())
def test_sl_golden(case, split):
    g = load_golden("sl.npz")
    inp, ref, edge = golden_case(g, case)
    if split:
        (r,), e = gpu_sl(**inp, split=True)
        if edge is not None:
            d = ref["layers_done"]
            assert_costs(e[0, 0, :, 0], edge[0, :, 0])
            assert_costs(e[0, 1:d], edge[1:d])
    else:
        (r,) = gpu_sl(**inp)
    assert_sl_parity(r, ref, edge)
    # untouched rows keep the reference's initial values
    d = ref["layers_done"]
    assert (r["cost_map"][d:] == 0).all() and (r["index_map"][d:] == -2).all()
    assert (r["index_map"][0] == -1).all()
    if not ref["found"]:
        assert (r["path_idx"] == -1).all()


def test_sl_pipeline_goldens():
    g = load_golden("pipe.npz")
    for case in [str(c) for c in g["cases"]]:
        p = case + "_"
        n_l = g[p + "l_map"].shape[1]
        (r,) = gpu_sl(g[p + "s_map"][None], g[p + "l_map"][None], int((n_l - 1) / 2), g[p + "ego"][None],
                      [g[p + "ob"]], np.array([float(g[p + "ob_dist"])]))
        assert r["found"] == bool(g[p + "found"])
        d = r["layers_done"]
        assert_costs(r["cost_map"][:d], g[p + "cost_map"][:d])
        assert np.array_equal(r["index_map"][:d], g[p + "index_map"][:d])
        if r["found"]:
            assert np.array_equal(r["path_idx"], g[p + "path_idx"])


def random_batch(rng, B, n_s, n_l, gap, n_ob, r_lo, r_hi, s_spread=None, jitter=False):
    s_layers = np.cumsum(np.full((B, n_s), gap), axis=1)
    if jitter:
        s_layers[:, -1] += rng.uniform(-0.4 * gap, 0.9 * gap, B)
    l_row = np.linspace(-1.45, 1.45, n_l)
    s_map = np.repeat(s_layers[:, :, None], n_l, axis=2)
    l_map = np.broadcast_to(l_row, (B, n_s, n_l)).copy()
    ego = np.stack([np.zeros(B), rng.uniform(-0.3, 0.3, B)], 1)
    hi = s_spread or s_layers[0, -1]
    obs = [np.stack([rng.uniform(0.5, hi, n_ob), rng.uniform(-2.5, 2.5, n_ob)], 1) for _ in range(B)]
    ob_dist = rng.uniform(r_lo, r_hi, B)
    return s_map, l_map, ego, obs, ob_dist


RANDOM_SHAPES = [
    # B, n_s, n_l, gap, n_ob, r_lo, r_hi, s_spread, jitter
    (24, 6, 5, 8.0, 3, 0.5, 1.0, None, True),
    (16, 12, 5, 8.0, 6, 0.3, 0.8, None, True),
    (12, 8, 16, 8.0, 10, 0.3, 0.8, None, False),
    (8, 10, 33, 1.5, 12, 0.2, 0.6, None, False),
    (6, 12, 64, 1.5, 20, 0.1, 0.4, None, False),
    (4, 9, 128, 1.5, 30, 0.1, 0.3, None, True),
    (3, 5, 200, 3.0, 10, 0.2, 0.5, None, False),
    (8, 6, 12, 8.0, 40, 0.4, 0.9, 12.0, False),    # many pairs per layer: mask path
    (4, 4, 16, 8.0, 300, 0.8, 1.2, 10.0, False),   # pair list overflows its LDS capacity
    (6, 5, 20, 8.0, 300, 0.8, 1.2, 14.0, False),   # the same on the thread-per-node kernel
    (16, 12, 5, 8.0, 900, 0.5, 0.9, None, True),   # default lattice, crowded scene (14 rounds of 64 obstacles)
    (8, 7, 2, 8.0, 20, 0.5, 0.9, None, True),      # lane-per-edge kernel, groups of 2 lanes
    (8, 7, 3, 8.0, 20, 0.5, 0.9, None, True),      # groups of 4
    (8, 9, 8, 4.0, 70, 0.4, 0.8, None, True),      # groups of 8, full
    (6, 9, 9, 4.0, 70, 0.4, 0.8, None, True),      # groups of 16, three waves
    (4, 30, 7, 1.5, 100, 0.2, 0.5, None, False),   # thin layers: 4 samples per edge
    (4, 3, 5, 30.0, 40, 0.5, 0.9, None, True),     # 61 samples per edge
]


@pytest.mark.parametrize("shape", RANDOM_SHAPES, ids=lambda s: "B%d_%dx%d_ob%d" % (s[0], s[1], s[2], s[4]))
@pytest.mark.parametrize("early_exit", [True, False])
def test_sl_random_vs_oracle(oracle, shape, early_exit):
    B, n_s, n_l, gap, n_ob, r_lo, r_hi, spread, jitter = shape
    rng = np.random.default_rng(1000 + n_s * n_l + n_ob)
    s_map, l_map, ego, obs, ob_dist = random_batch(rng, B, n_s, n_l, gap, n_ob, r_lo, r_hi, spread, jitter)
    mid_l = int((n_l - 1) / 2)
    res = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early_exit)
    res_split, edges = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early_exit, split=True)
    res_row = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early_exit, shared_row=True)
    others = [res_split, res_row]
    if n_l <= 16:
        others.append(gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early_exit, per_node=True))
    n_found = 0
    for b in range(B):
        small = n_s * n_l * n_l <= 40000
        ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b],
                                  no_early_exit=not early_exit, want_edges=small)
        assert_sl_parity(res[b], ref, ref.get("edge_cost"))
        n_found += ref["found"]
        # fused == edge tensor + DP kernel == shared-row layout (== thread-per-node kernel), bit for bit
        for other in (o[b] for o in others):
            assert np.array_equal(res[b]["cost_map"], other["cost_map"])
            assert np.array_equal(res[b]["index_map"], other["index_map"])
            assert np.array_equal(res[b]["path_idx"], other["path_idx"])
            assert res[b]["status"] == other["status"] and res[b]["layers_done"] == other["layers_done"]
        if small:
            d = ref["layers_done"]
            assert_costs(edges[b, 0, :, 0], ref["edge_cost"][0, :, 0])
            assert_costs(edges[b, 1:d], ref["edge_cost"][1:d])
    print("found %d of %d" % (n_found, B))


@pytest.mark.parametrize("n_l", [5, 12, 40])
def test_sl_layer_count_per_scenario(oracle, n_l):
    """Scenarios of one batch with different layer counts (SLMap derives the count from
    the reference line's length): each must equal the same scenario planned alone, and
    the unused tail of every output row keeps the initial values."""
    from motion_planning_amd import _lib
    rng = np.random.default_rng(77 + n_l)
    B, n_sm = 12, 11
    s_map, l_map, ego, obs, ob_dist = random_batch(rng, B, n_sm, n_l, 8.0, 14, 0.4, 0.9, None, False)
    n_layers = rng.integers(1, n_sm + 1, B).astype(np.int32)
    n_layers[0], n_layers[1], n_layers[2] = n_sm, 0, 1
    mid_l = int((n_l - 1) / 2)
    for early in (True, False):
        res = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early, n_layers=n_layers)
        res_split, _ = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early, n_layers=n_layers, split=True)
        for b in range(B):
            n = int(n_layers[b])
            for r in (res[b], res_split[b]):
                if n == 0:
                    assert r["status"] == _lib.MP_SL_SKIPPED and r["layers_done"] == 0
                    assert (r["path_idx"] == -1).all() and (r["index_map"] == -2).all() and (r["cost_map"] == 0).all()
                    continue
                ref = oracle.sl_find_path(s_map[b, :n], l_map[b, :n], mid_l, ego[b], obs[b], ob_dist[b],
                                          no_early_exit=not early, want_edges=True)
                got = dict(r, cost_map=r["cost_map"][:n], index_map=r["index_map"][:n], path_idx=r["path_idx"][:n])
                assert_sl_parity(got, ref, ref.get("edge_cost"))
                assert (r["cost_map"][n:] == 0).all() and (r["index_map"][n:] == -2).all()
                assert (r["path_idx"][n:] == -1).all()
            assert np.array_equal(res[b]["cost_map"], res_split[b]["cost_map"])
            assert np.array_equal(res[b]["index_map"], res_split[b]["index_map"])
            assert np.array_equal(res[b]["path_idx"], res_split[b]["path_idx"])


@pytest.mark.parametrize("n_l", [5, 12, 24])
@pytest.mark.parametrize("cap", [2, 8, 40])
def test_sl_pair_store_refill(oracle, cap, n_l, knob):
    """Both fused kernels keep the (sample, obstacle) pairs of all layers in an LDS store
    sized from the obstacle count; when it runs full the remaining layers are laid out
    again, and a layer too big for the whole store tests its pairs directly.  A tiny
    store forces both; results must not change."""
    knob("MP_SL_PAIR_CAP", cap)
    rng = np.random.default_rng(5 + cap)
    B, n_s = 6, 14
    s_map, l_map, ego, obs, ob_dist = random_batch(rng, B, n_s, n_l, 2.0, 30, 0.3, 0.9, None, True)
    mid_l = int((n_l - 1) / 2)
    for early in (True, False):
        res = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early)
        res_split, _ = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early, split=True)
        for b in range(B):
            ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b],
                                      no_early_exit=not early, want_edges=True)
            assert_sl_parity(res[b], ref, ref.get("edge_cost"))
            assert np.array_equal(res[b]["cost_map"], res_split[b]["cost_map"])
            assert np.array_equal(res[b]["index_map"], res_split[b]["index_map"])


@pytest.mark.parametrize("n_l,n_pairs", [(33, 150), (65, 40), (9, 60)])
def test_sl_mirror_symmetric_ties(oracle, n_l, n_pairs):
    """A lattice, an ego and obstacles that are mirror images about l = 0 make the costs
    of lanes t and n_l-1-t equal to the last bit on every layer, blocked or free: every
    minimum is a tie and the lower index must win, whatever order the kernel visits the
    predecessors in (crowded layers take the bound-first walk)."""
    rng = np.random.default_rng(900 + n_l)
    B, n_s = 6, 6
    s_map, l_map, ego, _, ob_dist = random_batch(rng, B, n_s, n_l, 8.0, 1, 0.4, 0.9, None, False)
    ego[:, 1] = 0.0
    l_map = (l_map - l_map[:, :, ::-1]) / 2.0   # exactly antisymmetric lateral rows
    obs = []
    for b in range(B):
        n = max(4, n_pairs >> (2 * (b % 3)))   # crowded, moderate and sparse scenes
        half = np.stack([rng.uniform(0.5, s_map[b, 2 + b % 3, 0], n), rng.uniform(0.05, 2.5, n)], 1)
        obs.append(np.concatenate([half, half * np.array([1.0, -1.0])]))
    mid_l = int((n_l - 1) / 2)
    n_ties = 0
    for early in (True, False):
        res = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early)
        res_split, _ = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early, split=True)
        for b in range(B):
            ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b],
                                      no_early_exit=not early, want_edges=True)
            assert_sl_parity(res[b], ref, ref.get("edge_cost"))
            assert np.array_equal(res[b]["index_map"], res_split[b]["index_map"])
            d = ref["layers_done"]
            cm = ref["cost_map"][:d]
            assert np.array_equal(cm, cm[:, ::-1])      # the symmetry is exact
            n_ties += d
    assert n_ties > 0


@pytest.mark.parametrize("n_l", [5, 40])
def test_sl_bad_scenarios_do_not_disturb_their_neighbours(n_l):
    """NaN / infinite obstacle coordinates, a NaN ego offset, NaN radius and counts outside
    the arrays in some scenarios of a batch: the call returns and the other scenarios
    come out bit for bit as in a clean batch."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(300 + n_l)
    B, n_s = 10, 9
    s_map, l_map, ego, obs, ob_dist = random_batch(rng, B, n_s, n_l, 8.0, 12, 0.4, 0.9, None, True)
    ob_s, ob_l, n_ob = pad_obstacles(obs)
    prm = mb.sl_params(early_exit=False, mid_l=int((n_l - 1) / 2))

    def run(ego_, ob_s_, ob_l_, n_ob_, dist_):
        sb = mb.SLBatch(s_map[:, :, 0], l_map, ego_, ob_s_, ob_l_, n_ob_, dist_, check=False)
        out = mb.sl_plan(sb, prm)
        torch.cuda.synchronize()
        return out

    clean = run(ego, ob_s, ob_l, n_ob, ob_dist)
    e2, s2, l2, n2, d2 = ego.copy(), ob_s.copy(), ob_l.copy(), n_ob.copy(), ob_dist.copy()
    s2[1, 0] = np.nan
    l2[2, 3] = np.inf
    e2[3, 1] = np.nan
    d2[4] = np.nan
    n2[5], n2[6] = -7, 10 ** 6
    s2[7, :] = -np.inf
    dirty = run(e2, s2, l2, n2, d2)
    good = [0, 8, 9]
    for k in ("cost_map", "index_map", "path_idx", "status", "layers_done"):
        assert torch.equal(getattr(clean, k)[good], getattr(dirty, k)[good]), k
    # counts are clamped to the scenario's own slots: scenario 5 plans without obstacles,
    # scenario 6 with all of its slots
    st = dirty.status.cpu().numpy()
    assert st[5] in (0, 1) and st[6] in (0, 1)


def test_sl_empty_edge_and_capacity_status():
    from motion_planning_amd import _lib
    s_map = np.array([[[8.0] * 5, [8.2] * 5]])            # second gap shorter than d_s
    l_map = np.broadcast_to(np.linspace(-1.45, 1.45, 5), (1, 2, 5)).copy()
    (r,) = gpu_sl(s_map, l_map, 2, np.zeros((1, 2)), [np.array([[3.0, 0.0]])], np.array([0.5]))
    assert r["status"] == _lib.MP_SL_E_EMPTY_EDGE
    s_map = np.array([[[80.0] * 5, [90.0] * 5]])          # 161 samples on the ego edge
    (r,) = gpu_sl(s_map, l_map, 2, np.zeros((1, 2)), [np.array([[3.0, 0.0]])], np.array([0.5]))
    assert r["status"] == _lib.MP_SL_E_CAPACITY
    (r,) = gpu_sl(np.array([[[8.0] * 5, [8.2] * 5]]), l_map, 2, np.zeros((1, 2)), [np.zeros((0, 2))],
                  np.array([0.0]))
    assert r["status"] == _lib.MP_SL_FOUND                  # no obstacles: empty edge costs 0


@pytest.mark.parametrize("n_l", [24, 128])
def test_sl_targets_that_no_edge_enters_without_a_collision(oracle, n_l):
    """The reference evaluates an edge quintic at the absolute arc length (path_planner.py:146-148),
    so the first sample of the second layer sits at the TARGET's lateral offset whatever the
    source: an obstacle next to it blocks every edge into the targets it reaches, and their
    costs all carry the penalty.  Also an obstacle on a node of a later layer."""
    rng = np.random.default_rng(31 + n_l)
    B, n_s = 6, 7
    s_map, l_map, ego, _, _ = random_batch(rng, B, n_s, n_l, 4.0, 1, 0.3, 0.8)
    obs = []
    for b in range(B):
        o = [[s_map[b, 0, 0] + rng.uniform(-0.1, 0.1), rng.uniform(-1.0, 1.0)],        # first sample of layer 1
             [s_map[b, 2, 0], rng.uniform(-1.0, 1.0)], [s_map[b, 4, 0] - 0.2, 0.3]]    # on / next to later nodes
        obs.append(np.array(o[: 1 + b % 3]))
    ob_dist = rng.uniform(0.3, 0.9, B)
    mid_l = int((n_l - 1) / 2)
    res = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=False)
    res_split, edges = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=False, split=True)
    n_dead = 0
    for b in range(B):
        ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b], no_early_exit=True,
                                  want_edges=True)
        assert_sl_parity(res[b], ref, ref["edge_cost"])
        assert np.array_equal(res[b]["cost_map"], res_split[b]["cost_map"])
        assert np.array_equal(res[b]["index_map"], res_split[b]["index_map"])
        n_dead += int((ref["edge_cost"][1] >= 3e4).all(axis=1).sum())     # targets of layer 1 whose every edge is penalised
    assert n_dead > 0


def test_sl_heaviest_first_order_changes_nothing(oracle, knob):
    """mp_sl_plan_ordered launches the workgroups heaviest scenario first; outputs must be
    those of mp_sl_plan bit for bit, also with per-scenario layer counts, scenarios without
    obstacles and a dirty workspace."""
    import torch
    from motion_planning_amd import batch as mb
    rng = np.random.default_rng(77)
    B, n_s, n_l = 1700, 9, 40        # (batches that fit on the chip at once are not ordered)
    s_map, l_map, ego, obs, ob_dist = random_batch(rng, B, n_s, n_l, 1.5, 14, 0.2, 0.7, jitter=True)
    ob_s, ob_l, n_ob = pad_obstacles(obs)
    n_ob[::7] = 0
    n_layers = rng.integers(0, n_s + 1, B).astype(np.int32)
    sb = mb.SLBatch(s_map[:, :, 0], l_map, ego, ob_s, ob_l, n_ob, ob_dist, n_layers=n_layers)
    prm = mb.sl_params(early_exit=True, n_l=n_l)
    plain = mb.sl_plan(sb, prm, ordered=False)
    out = mb.SLPlanOut(B, n_s, n_l)
    out.order_ws.fill_(-12345)
    ordered = mb.sl_plan(sb, prm, out)
    torch.cuda.synchronize()
    order = out.order_ws[B:].cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(B))                 # a permutation of the scenarios
    key = out.order_ws[:B].cpu().numpy()
    assert (np.diff(key[order]) <= 0).all() and key.max() > key.min()   # heaviest bin first
    for k in ("cost_map", "index_map", "path_idx", "status", "layers_done"):
        assert torch.equal(getattr(plain, k), getattr(ordered, k)), k
    assert np.array_equal(plain.best_cost.cpu().numpy(), ordered.best_cost.cpu().numpy(), equal_nan=True)
    b = int(order[0])                                                   # the scenario rated heaviest, against the oracle
    if n_layers[b] > 0:
        L = n_layers[b]
        ref = oracle.sl_find_path(s_map[b, :L], l_map[b, :L], int((n_l - 1) / 2), ego[b], obs[b][: n_ob[b]], ob_dist[b])
        got = dict(found=int(ordered.status[b]) == 1, cost_map=ordered.cost_map[b, :L].cpu().numpy(),
                   index_map=ordered.index_map[b, :L].cpu().numpy(), path_idx=ordered.path_idx[b, :L].cpu().numpy(),
                   layers_done=int(ordered.layers_done[b]))
        assert_sl_parity(got, ref)
    # a batch that fits on the chip at once is launched as it comes, and so is any batch with the knob off
    small = mb.SLBatch(s_map[:64, :, 0], l_map[:64], ego[:64], ob_s[:64], ob_l[:64], n_ob[:64], ob_dist[:64],
                       n_layers=n_layers[:64])
    out_s = mb.SLPlanOut(64, n_s, n_l)
    out_s.order_ws.fill_(-12345)
    mb.sl_plan(small, prm, out_s)
    knob("MP_ORDER", 0)
    out.order_ws.fill_(-12345)
    off = mb.sl_plan(sb, prm, out)
    torch.cuda.synchronize()
    assert (out_s.order_ws == -12345).all() and (out.order_ws == -12345).all()
    assert torch.equal(off.path_idx, plain.path_idx) and torch.equal(out_s.path_idx, plain.path_idx[:64])


def gpu_sl_nodes(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=True, n_layers=None):
    """Per-node edge kernel + DP kernel; returns (list of result dicts, edge tensor, edge status)."""
    import torch
    from motion_planning_amd import batch as mb
    ob_s, ob_l, n_ob = pad_obstacles(obs)
    prm = mb.sl_params(early_exit=early_exit, mid_l=mid_l)
    edge, st = mb.sl_edge_cost_nodes(s_map, l_map, ego, ob_s, ob_l, n_ob, ob_dist, prm, n_layers=n_layers)
    out = mb.sl_dp(edge, prm, n_layers=n_layers)
    torch.cuda.synchronize()
    res = [dict(found=int(out.status[b]) == 1, status=int(out.status[b]), cost_map=out.cost_map[b].cpu().numpy(),
                index_map=out.index_map[b].cpu().numpy(), path_idx=out.path_idx[b].cpu().numpy(),
                layers_done=int(out.layers_done[b])) for b in range(len(s_map))]
    return res, edge.cpu().numpy(), st.cpu().numpy()


@pytest.mark.parametrize("shape", [(6, 5, 5, 8.0, 6), (4, 6, 12, 3.0, 15), (2, 3, 40, 20.0, 30), (3, 3, 5, 90.0, 8)],
                         ids=lambda s: "B%d_%dx%d_gap%g" % s[:4])
def test_sl_per_node_lattice_vs_oracle(oracle, shape):
    """Every node at its own arc length (the reference reads s_map[s][l] per node,
    path_planner.py:106-109); the widest gaps give edges of more than 128 samples."""
    B, n_s, n_l, gap, n_ob = shape
    rng = np.random.default_rng(4000 + n_s * n_l)
    s_map, l_map, ego, obs, ob_dist = random_batch(rng, B, n_s, n_l, gap, n_ob, 0.3, 0.8)
    s_map = s_map + rng.uniform(-0.2 * gap, 0.2 * gap, s_map.shape)
    l_map = l_map + rng.uniform(-0.05, 0.05, l_map.shape)
    mid_l = int((n_l - 1) / 2)
    res, edge, st = gpu_sl_nodes(s_map, l_map, mid_l, ego, obs, ob_dist)
    assert (st == 1).all()
    for b in range(B):
        ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b], no_early_exit=True,
                                  want_edges=True)
        assert_costs(edge[b, 0, :, 0], ref["edge_cost"][0, :, 0])
        assert_costs(edge[b, 1:], ref["edge_cost"][1:])
        assert np.isnan(edge[b, 0, :, 1:]).all()
        ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b])
        assert_sl_parity(res[b], ref)


def test_sl_per_node_kernel_equals_layer_kernel_on_layered_lattices():
    rng = np.random.default_rng(5)
    s_map, l_map, ego, obs, ob_dist = random_batch(rng, 5, 6, 9, 6.0, 12, 0.3, 0.8, jitter=True)
    n_layers = np.array([6, 3, 1, 6, 4], np.int32)
    res, edge, st = gpu_sl_nodes(s_map, l_map, 4, ego, obs, ob_dist, n_layers=n_layers)
    res_l, edge_l = gpu_sl(s_map, l_map, 4, ego, obs, ob_dist, split=True, n_layers=n_layers)
    for b in range(5):
        d = n_layers[b]
        assert_costs(edge[b, 0, :, 0], edge_l[b, 0, :, 0])
        assert_costs(edge[b, 1:d], edge_l[b, 1:d])
        assert np.isnan(edge[b, d:]).all()
        assert res[b]["status"] == res_l[b]["status"] and res[b]["layers_done"] == res_l[b]["layers_done"]
        assert np.array_equal(res[b]["index_map"], res_l[b]["index_map"])


def test_sl_per_node_empty_edge_status():
    from motion_planning_amd import _lib
    s_map = np.array([[[8.0, 8.0, 8.0], [16.0, 8.3, 16.0]]])     # node (1,1) 0.3 m behind layer 0: no sample
    l_map = np.broadcast_to(np.linspace(-1, 1, 3), (1, 2, 3)).copy()
    _, edge, st = gpu_sl_nodes(s_map, l_map, 1, np.zeros((1, 2)), [np.array([[3.0, 0.0]])], np.array([0.5]))
    assert st[0] == _lib.MP_SL_E_EMPTY_EDGE and np.isnan(edge[0, 1, 1]).all() and np.isfinite(edge[0, 1, 0]).all()
    _, edge, st = gpu_sl_nodes(s_map, l_map, 1, np.zeros((1, 2)), [np.zeros((0, 2))], np.array([0.5]))
    assert st[0] == _lib.MP_SL_FOUND and (edge[0, 1, 1] == 0).all()     # no obstacles: the empty edge costs 0


@pytest.mark.parametrize("n_l", [5, 24])
def test_sl_layers_out_of_order(oracle, n_l):
    """A layer placed behind its predecessor ends the plan there (an edge without samples
    next to obstacles, the reference's ValueError), but the layers in front of it are
    relaxed as usual -- with their obstacle pairs found although the layers' arc lengths
    do not ascend."""
    from motion_planning_amd import _lib
    rng = np.random.default_rng(8 + n_l)
    B, n_s = 4, 6
    s_map, l_map, ego, obs, ob_dist = random_batch(rng, B, n_s, n_l, 8.0, 25, 0.4, 0.9, None, False)
    s_map[1, 3] = 20.0      # layer 3 of scenario 1 lies behind layer 2 (24 m)
    s_map[2, 1] = 4.0       # layer 1 of scenario 2 lies behind layer 0 (8 m)
    mid_l = int((n_l - 1) / 2)
    res = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=False)
    # the other fused kernel on the same lattice (lane-per-edge <-> thread-per-node)
    res_other = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=False, per_node=True) if n_l <= 16 else res
    for b, bad in ((1, 3), (2, 1)):
        assert res[b]["status"] == _lib.MP_SL_E_EMPTY_EDGE and res[b]["layers_done"] == bad
        ref = oracle.sl_find_path(s_map[b, :bad], l_map[b, :bad], mid_l, ego[b], obs[b], ob_dist[b],
                                  no_early_exit=True)
        assert_costs(res[b]["cost_map"][:bad], ref["cost_map"])
        assert np.array_equal(res[b]["index_map"][:bad], ref["index_map"])
        assert (res[b]["index_map"][bad:] == -2).all() and (res[b]["path_idx"] == -1).all()
    for b in (0, 3):
        ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b], no_early_exit=True)
        assert_sl_parity(res[b], ref, None)
    for b in range(B):
        assert np.array_equal(res[b]["index_map"], res_other[b]["index_map"])
        assert np.array_equal(res[b]["cost_map"], res_other[b]["cost_map"]) and res[b]["status"] == res_other[b]["status"]


@pytest.mark.parametrize("n_l", [32, 100])
@pytest.mark.parametrize("early_exit", [True, False])
def test_sl_wall_far_down_the_lattice(oracle, n_l, early_exit):
    """Walls of obstacles across every lane, deep in the lattice where a shared lateral
    row lets the fused kernel relax layers without exchanging nodes or minima: four
    blocked layers in a row exceed the abort threshold.  Cost map, index map, status and
    the number of layers relaxed must match the oracle with and without early exit, and
    the per-layer-row layout (which always takes the walk)."""
    B, n_s = 6, 24
    rng = np.random.default_rng(61 + n_l)
    s_map, l_map, ego, _, ob_dist = random_batch(rng, B, n_s, n_l, 1.5, 1, 0.45, 0.55, None, False)
    obs = []
    for b in range(B):
        first = 10 + 2 * b                      # first walled layer
        n_wall = [4, 3, 5, 4, 1, 4][b]          # 3 walls stay under the threshold, 4 or more do not
        wall_l = np.arange(-1.8, 1.81, 0.3)
        pts = [(1.5 * (k + 1) - 0.05, l) for k in range(first, min(first + n_wall, n_s)) for l in wall_l]
        pts += [(rng.uniform(1, 12), rng.uniform(-2, 2)) for _ in range(3)]
        obs.append(np.array(pts))
    mid_l = int((n_l - 1) / 2)
    res = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early_exit, shared_row=True)
    res_rows = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early_exit)
    n_found = 0
    for b in range(B):
        ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b],
                                  no_early_exit=not early_exit)
        assert_sl_parity(res[b], ref, None)
        assert res[b]["layers_done"] == ref["layers_done"], b
        n_found += ref["found"]
        for k in ("cost_map", "index_map", "path_idx"):
            assert np.array_equal(res[b][k], res_rows[b][k]), (b, k)
        assert res[b]["status"] == res_rows[b]["status"] and res[b]["layers_done"] == res_rows[b]["layers_done"]
    assert 0 < n_found < B


def test_sl_wall_then_short_gap_reports_what_the_reference_reaches_first(oracle):
    """A wall the reference gives up at (four blocked layers), followed further down by a
    layer gap shorter than the sample spacing (where the reference would raise, min of an
    empty array).  With early exit the reference never reaches the short gap, so the
    scenario is "not found" with the layers up to the wall relaxed; scenarios whose wall
    stays under the threshold do reach the gap and report the error."""
    B, n_s, n_l = 4, 30, 64
    rng = np.random.default_rng(77)
    s_map, l_map, ego, _, ob_dist = random_batch(rng, B, n_s, n_l, 1.5, 1, 0.45, 0.55, None, False)
    obs = []
    for b in range(B):
        first = 9 + 2 * b
        n_wall = [4, 2, 5, 3][b]
        wall_l = np.arange(-1.8, 1.81, 0.3)
        pts = [(1.5 * (k + 1) - 0.05, l) for k in range(first, first + n_wall) for l in wall_l]
        obs.append(np.array(pts))
    s_map = s_map.copy()
    s_map[:, 24:, :] = s_map[:, 23:24, :] + 0.2 + 1.5 * np.arange(n_s - 24)[None, :, None]   # layer 24 sits 0.2 m behind layer 23
    mid_l = int((n_l - 1) / 2)
    res = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=True, shared_row=True)
    res_rows = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=True)
    kinds = set()
    for b in range(B):
        try:
            ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b])
        except ValueError:
            ref = None
        kinds.add(ref is None)
        for r in (res[b], res_rows[b]):
            if ref is None:
                assert r["status"] == -2, (b, r["status"])        # MP_SL_E_EMPTY_EDGE
            else:
                assert not ref["found"] and r["status"] == 0, (b, r["status"])
                assert r["layers_done"] == ref["layers_done"], b
                n = ref["layers_done"]
                assert_costs(r["cost_map"][:n], ref["cost_map"][:n])
                assert (r["index_map"][n:] == -2).all() and (r["cost_map"][n:] == 0).all()
    assert kinds == {True, False}


def test_sl_largest_fused_lattice(oracle):
    """256 layers x 256 lanes, the largest lattice the fused kernel takes (its predecessor
    table alone is 64 KB of LDS): fused == edge tensor + DP kernel bit for bit, and the first
    layers against the oracle."""
    rng = np.random.default_rng(41)
    B, n_s, n_l = 2, 256, 256
    s_map, l_map, ego, obs, ob_dist = random_batch(rng, B, n_s, n_l, 0.5, 40, 0.2, 0.5, None, False)
    mid_l = int((n_l - 1) / 2)
    res = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=False)
    res_split, _ = gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=False, split=True)
    for b in range(B):
        assert res[b]["layers_done"] == n_s and res[b]["status"] in (0, 1)
        assert np.array_equal(res[b]["cost_map"], res_split[b]["cost_map"])
        assert np.array_equal(res[b]["index_map"], res_split[b]["index_map"])
        assert np.array_equal(res[b]["path_idx"], res_split[b]["path_idx"])
        n = 6
        ref = oracle.sl_find_path(s_map[b, :n], l_map[b, :n], mid_l, ego[b], obs[b], ob_dist[b], no_early_exit=True,
                                  want_edges=True)
        assert_costs(res[b]["cost_map"][:n], ref["cost_map"])
        bad = np.argwhere(res[b]["index_map"][:n] != ref["index_map"])
        for s_, l_ in bad:   # only exact ties may differ
            cand = ref["edge_cost"][s_, l_, :] + ref["cost_map"][s_ - 1]
            a_, b_ = cand[res[b]["index_map"][s_, l_]], cand[ref["index_map"][s_, l_]]
            assert abs(a_ - b_) <= 1e-9 * max(abs(a_), abs(b_))


def test_sl_unsupported_size_is_an_error():
    from motion_planning_amd import batch as mb, _lib
    B, n_s, n_l = 1, 2, 300
    sb = mb.SLBatch(np.array([[8.0, 16.0]]), np.linspace(-1, 1, n_l)[None], np.zeros((1, 2)),
                    np.zeros((1, 1)), np.zeros((1, 1)), np.zeros(1, np.int32), np.zeros(1))
    with pytest.raises(_lib.MpError):
        mb.sl_plan(sb)
    # the edge-tensor path takes any lattice width
    edge, st = mb.sl_edge_cost(sb)
    out = mb.sl_dp(edge, mb.sl_params(n_l=n_l))
    assert int(out.status[0]) == 1


def cfg3_batch(B, seed0=1000, n_s=64, n_l=128, n_ob=50):
    """Batched SL workload: 64 layers x 128 lanes, 1.5 m layer gap, 50 obstacles."""
    s_layers = np.broadcast_to(1.5 * (np.arange(n_s) + 1.0), (B, n_s)).copy()
    l_row = np.broadcast_to(np.linspace(-1.45, 1.45, n_l), (B, n_l)).copy()
    ego = np.zeros((B, 2))
    ob_s = np.zeros((B, n_ob))
    ob_l = np.zeros((B, n_ob))
    ob_dist = np.zeros(B)
    for b in range(B):
        rng = np.random.Generator(np.random.PCG64(seed0 + b))
        ego[b, 1] = rng.uniform(-0.3, 0.3)
        ob_s[b] = rng.uniform(2, 96, n_ob)
        ob_l[b] = rng.uniform(-2.5, 2.5, n_ob)
        ob_dist[b] = rng.uniform(0.3, 0.8)
    return s_layers, l_row, ego, ob_s, ob_l, np.full(B, n_ob, np.int32), ob_dist


def test_sl_cfg3_subset_vs_oracle_and_properties(oracle):
    import torch
    from motion_planning_amd import batch as mb
    B = 4096
    s_layers, l_row, ego, ob_s, ob_l, n_ob, ob_dist = cfg3_batch(B)
    sb = mb.SLBatch(s_layers, l_row, ego, ob_s, ob_l, n_ob, ob_dist)
    for early in (True, False):
        out = mb.sl_plan(sb, mb.sl_params(early_exit=early, n_l=128))
        out2 = mb.sl_plan(sb, mb.sl_params(early_exit=early, n_l=128))
        torch.cuda.synchronize()
        # run-to-run identical
        assert torch.equal(out.cost_map, out2.cost_map) and torch.equal(out.index_map, out2.index_map)
        status = out.status.cpu().numpy()
        assert set(np.unique(status)) <= {0, 1}
        idx = out.index_map.cpu().numpy()
        cm = out.cost_map.cpu().numpy()
        path = out.path_idx.cpu().numpy()
        done = out.layers_done.cpu().numpy()
        if not early:
            assert (done == 64).all()
        found = np.flatnonzero(status == 1)
        for b in found[:200]:
            # backtrack consistency and end node = first argmin of the last row
            assert path[b, -1] == int(np.argmin(cm[b, -1]))
            for s in range(63, 0, -1):
                assert path[b, s - 1] == idx[b, s, path[b, s]]
            assert float(out.best_cost[b]) == cm[b, -1].min()
        for b in np.flatnonzero(status == 0)[:200]:
            assert (path[b] == -1).all()
            assert cm[b, done[b] - 1].min() > 1e5 if early else True
        # the instance that writes no maps (the one the benchmark times) decides the same plan
        lean = mb.sl_plan(sb, mb.sl_params(early_exit=early, n_l=128), want_maps=False)
        torch.cuda.synchronize()
        assert lean.cost_map is None and lean.index_map is None
        assert torch.equal(lean.path_idx, out.path_idx) and torch.equal(lean.status, out.status)
        assert torch.equal(lean.layers_done, out.layers_done)
        assert np.array_equal(lean.best_cost.cpu().numpy(), out.best_cost.cpu().numpy(), equal_nan=True)
        # full-size scenarios against the oracle, found and not found alike
        n_each = 32 if early else 8
        sample = list(found[:n_each]) + list(np.flatnonzero(status == 0)[:n_each])
        assert len(sample) == 2 * n_each

        def reference(b):
            s_map = np.repeat(s_layers[b][:, None], 128, axis=1)
            l_map = np.broadcast_to(l_row[b], (64, 128))
            return oracle.sl_find_path(s_map, l_map, 63, ego[b], np.stack([ob_s[b], ob_l[b]], 1), ob_dist[b],
                                       no_early_exit=not early)

        with ThreadPoolExecutor(8) as pool:      # the oracle is a C call that releases the interpreter lock
            refs = list(pool.map(reference, sample))
        for b, ref in zip(sample, refs):
            got = dict(found=status[b] == 1, cost_map=cm[b], index_map=idx[b], path_idx=path[b],
                       layers_done=done[b])
            assert_sl_parity(got, ref)
        print("cfg3 early_exit=%s: found %d / %d" % (early, len(found), B))


def test_lattice_step_in_two_parts_on_two_streams_equals_one_stream():
    """The benchmark's headline schedule: the batch cut into two parts, part k running SL plan
    -> ST DP on stream k for several steps with no part waiting for the other.  Same plans,
    bit for bit, as the whole batch on one stream."""
    import torch
    import bench
    from motion_planning_amd import batch as mb
    B = 3001                                      # parts of 1501 and 1500: ordered SL launch on both
    sc = bench.make_scenarios(7, B)
    prm = mb.sl_params(early_exit=False, n_l=bench.N_L)

    def run(lo, hi, stream, steps):
        sb = mb.SLBatch(sc["s_layers"][lo:hi], sc["l_row"][lo:hi], sc["ego"][lo:hi], sc["ob_s"][lo:hi],
                        sc["ob_l"][lo:hi], sc["n_ob"][lo:hi], sc["ob_dist"][lo:hi])
        st_in = (mb.to_dev(sc["st_n_s"][lo:hi], torch.int32), mb.to_dev(sc["down"][lo:hi]),
                 mb.to_dev(sc["up"][lo:hi]), mb.to_dev(sc["n_reg"][lo:hi], torch.int32),
                 mb.to_dev(sc["cur_vel"][lo:hi]))
        out, st = mb.SLPlanOut(hi - lo, bench.N_S, bench.N_L, want_maps=False), None
        torch.cuda.synchronize()

        def go():
            nonlocal st
            with torch.cuda.stream(stream):
                for _ in range(steps):
                    mb.sl_plan(sb, prm, out)
                    st = mb.st_dp(*st_in, bench.SPEED_LIM, want_tables=False, n_s_max=bench.ST_N_S, out=st)
        return go, out, lambda: st

    whole, w_out, w_st = run(0, B, torch.cuda.current_stream(), 1)
    whole()
    torch.cuda.synchronize()
    parts = [run(0, 1501, torch.cuda.Stream(), 3), run(1501, B, torch.cuda.Stream(), 3)]
    for go, _, _ in parts:
        go()
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([o.path_idx for _, o, _ in parts]), w_out.path_idx)
    assert torch.equal(torch.cat([o.status for _, o, _ in parts]), w_out.status)
    assert torch.equal(torch.cat([o.layers_done for _, o, _ in parts]), w_out.layers_done)
    assert np.array_equal(torch.cat([o.best_cost for _, o, _ in parts]).cpu().numpy(), w_out.best_cost.cpu().numpy(),
                          equal_nan=True)
    for k in ("path_s_ind", "path_len"):
        assert torch.equal(torch.cat([s()[k] for _, _, s in parts]), w_st()[k]), k
    assert 0 < int((w_out.status == 1).sum()) < B


@pytest.mark.parametrize("rows", [None, 1, 2, 5])
@pytest.mark.parametrize("early_exit", [True, False])
def test_sl_lean_instance_with_few_predecessor_rows(rows, early_exit):
    """The instance that writes no maps keeps predecessor rows only for layers that looked at
    other lanes, a fixed number of them; a scenario in which more layers walk is planned again
    by a second launch that holds every row.  With the row count forced down (MP_SL_ROWS) most
    scenarios take that second launch: same plans, bit for bit, as the instance that writes the
    maps (which always holds every row), also where no layer runs without exchange (a penalty
    so large that the cost bound never falls below the lane spacing)."""
    import torch
    from motion_planning_amd import _lib, batch as mb
    shapes = [(8, 10, 33, 1.5, 12, 0.2, 0.6), (6, 12, 64, 1.5, 20, 0.1, 0.4), (4, 40, 128, 1.5, 30, 0.1, 0.3),
              (5, 30, 20, 1.5, 25, 0.2, 0.5), (3, 64, 128, 1.5, 50, 0.3, 0.8)]
    n_retry_shapes = 0
    for k, (B, n_s, n_l, gap, n_ob, r_lo, r_hi) in enumerate(shapes):
        rng = np.random.default_rng(300 + k)
        s_map, l_map, ego, obs, ob_dist = random_batch(rng, B, n_s, n_l, gap, n_ob, r_lo, r_hi, None, False)
        ob_s, ob_l, n_obs = pad_obstacles(obs)
        sb = mb.SLBatch(s_map[:, :, 0], l_map[:, 0, :], ego, ob_s, ob_l, n_obs, ob_dist)
        for no_path_cost in (1e4, 1e13):
            prm = mb.sl_params(early_exit=early_exit, n_l=n_l, no_path_cost=no_path_cost)
            full = mb.sl_plan(sb, prm, want_maps=True)
            _lib.set_knob("MP_SL_ROWS", -1 if rows is None else rows)
            try:
                lean = mb.sl_plan(sb, prm, want_maps=False)
                lean_unordered = mb.sl_plan(sb, prm, want_maps=False, ordered=False)
            finally:
                _lib.set_knob("MP_SL_ROWS", -1)
            torch.cuda.synchronize()
            for o in (lean, lean_unordered):
                assert torch.equal(o.status, full.status) and torch.equal(o.path_idx, full.path_idx)
                assert torch.equal(o.layers_done, full.layers_done)
                assert np.array_equal(o.best_cost.cpu().numpy(), full.best_cost.cpu().numpy(), equal_nan=True)
            assert set(np.unique(full.status.cpu().numpy())) <= {0, 1}
            # layers that walked in a found plan: where the path changes lane at all, a row was needed
            path = full.path_idx.cpu().numpy()
            found = full.status.cpu().numpy() == 1
            if found.any() and rows is not None:
                n_retry_shapes += int((np.count_nonzero(np.diff(path[found], axis=1), axis=1) > rows).any())
    if rows in (1, 2):
        assert n_retry_shapes > 0, "no scenario needed more rows than the forced count: the second launch never ran"


@pytest.mark.parametrize("cap", [None, 16, 128])
@pytest.mark.parametrize("n_ob", [2, 6, 20, 50])
def test_sl_few_pairs_in_long_runs_of_layers_without_exchange(oracle, knob, n_ob, cap):
    """On the benchmark's lattice all but the first few layers keep every lane and are taken
    in runs of up to 64, whose (sample, obstacle) pairs are spread over the lanes of a wave.
    With few obstacles a run has fewer pairs than layers, and with a small pair store a run
    can start right behind a refill with a handful of pairs in it: the lanes that took no pair
    still have to see what the others found for their layer.  Every instance against the
    oracle."""
    import torch
    from motion_planning_amd import batch as mb
    B = 24
    s_layers, l_row, ego, ob_s, ob_l, n_obs, ob_dist = cfg3_batch(B, seed0=5000 + n_ob, n_ob=n_ob)
    if n_ob == 50:
        # the scenario whose refill at 128 slots leaves a run of five layers with four pairs
        s_layers, l_row, ego, ob_s, ob_l, n_obs, ob_dist = cfg3_batch(B, seed0=1000 + 13691 - 3)
    else:
        ob_s[:, : n_ob // 2] = 30.0 + 0.66 * ob_s[:, : n_ob // 2]   # half of them well inside the runs
    sb = mb.SLBatch(s_layers, l_row, ego, ob_s, ob_l, n_obs, ob_dist)
    prm = mb.sl_params(early_exit=False, n_l=128)
    if cap is not None:
        knob("MP_SL_PAIR_CAP", cap)
    full = mb.sl_plan(sb, prm, want_maps=True)
    lean = mb.sl_plan(sb, prm, want_maps=False)
    lean_unordered = mb.sl_plan(sb, prm, want_maps=False, ordered=False)
    counted, _, _ = mb.sl_plan_counted(sb, prm)
    torch.cuda.synchronize()
    for o in (lean, lean_unordered, counted):
        assert torch.equal(o.status, full.status) and torch.equal(o.path_idx, full.path_idx)
        assert np.array_equal(o.best_cost.cpu().numpy(), full.best_cost.cpu().numpy(), equal_nan=True)

    def reference(b):
        s_map = np.repeat(s_layers[b][:, None], 128, axis=1)
        l_map = np.broadcast_to(l_row[b], (64, 128))
        return oracle.sl_find_path(s_map, l_map, 63, ego[b], np.stack([ob_s[b], ob_l[b]], 1), ob_dist[b],
                                   no_early_exit=True)

    with ThreadPoolExecutor(8) as pool:
        refs = list(pool.map(reference, range(B)))
    for b, ref in enumerate(refs):
        got = dict(found=int(full.status[b]) == 1, cost_map=full.cost_map[b].cpu().numpy(),
                   index_map=full.index_map[b].cpu().numpy(), path_idx=full.path_idx[b].cpu().numpy(),
                   layers_done=int(full.layers_done[b]))
        assert_sl_parity(got, ref)


FUZZ_SHAPES = [
    # B, n_s, n_l, gap, n_ob_max, r_lo, r_hi
    (1024, 64, 128, 1.5, 50, 0.3, 0.8),     # the benchmark's lattice with 0..50 obstacles
    (1024, 64, 128, 1.5, 8, 0.3, 0.9),      # sparse: runs of layers with fewer pairs than layers
    (2048, 40, 64, 2.0, 30, 0.2, 0.9),
    (2048, 20, 33, 3.0, 60, 0.2, 0.7),
    (2048, 64, 17, 1.5, 120, 0.3, 0.8),     # one wave of 17 lanes, crowded
    (4096, 12, 16, 8.0, 100, 0.3, 0.9),     # lane-per-edge kernel, full groups
    (4096, 12, 5, 8.0, 40, 0.3, 0.9),       # the default lattice
    (512, 10, 200, 1.5, 30, 0.1, 0.4),      # four waves per scenario
]


@pytest.mark.parametrize("shape", FUZZ_SHAPES, ids=lambda s: "B%d_%dx%d_ob%d" % (s[0], s[1], s[2], s[4]))
def test_sl_many_random_scenarios_fused_against_materialised(knob, shape):
    """Thousands of random scenarios per lattice shape (obstacle counts 0..max, layer counts that
    differ per scenario, both exit modes, the pair store at its own size and forced to refill):
    every fused instance against the materialised path (edge tensor + plain DP), bit for bit.
    The oracle pins both on small batches above; this looks for the scenario in ten thousand."""
    import torch
    from motion_planning_amd import batch as mb
    B, n_s, n_l, gap, n_ob_max, r_lo, r_hi = shape
    rng = np.random.default_rng(77 + n_s * n_l + n_ob_max)
    s_layers = np.cumsum(np.full((B, n_s), gap), axis=1)
    s_layers[:, -1] += rng.uniform(-0.4 * gap, 0.9 * gap, B)
    l_row = np.broadcast_to(np.linspace(-1.45, 1.45, n_l), (B, n_l)).copy()
    ego = np.stack([np.zeros(B), rng.uniform(-0.3, 0.3, B)], 1)
    ob_s = rng.uniform(0.5, s_layers[0, -1], (B, n_ob_max))
    ob_l = rng.uniform(-2.5, 2.5, (B, n_ob_max))
    n_ob = rng.integers(0, n_ob_max + 1, B).astype(np.int32)
    ob_dist = rng.uniform(r_lo, r_hi, B)
    n_layers = np.where(rng.random(B) < 0.25, rng.integers(1, n_s + 1, B), n_s).astype(np.int32)
    sb = mb.SLBatch(s_layers, l_row, ego, ob_s, ob_l, n_ob, ob_dist, n_layers=n_layers)
    n_found = 0
    for early_exit in (True, False):
        prm = mb.sl_params(early_exit=early_exit, n_l=n_l)
        edge, est = mb.sl_edge_cost(sb, prm)
        plain = mb.sl_dp(edge, prm, n_layers=n_layers)
        del edge
        torch.cuda.synchronize()
        assert bool((est == 1).all())
        for cap in (None, 32):
            if cap is not None:
                knob("MP_SL_PAIR_CAP", cap)
            outs = [mb.sl_plan(sb, prm, want_maps=True), mb.sl_plan(sb, prm, want_maps=False),
                    mb.sl_plan(sb, prm, want_maps=False, ordered=False)]
            if n_l <= 16:
                knob("MP_SL_NARROW", 0)     # the same lattice through the thread-per-node kernel
                outs.append(mb.sl_plan(sb, prm, want_maps=True))
                knob("MP_SL_NARROW", -1)
            knob("MP_SL_PAIR_CAP", -1)
            torch.cuda.synchronize()
            for o in outs:
                bad = ((o.path_idx != plain.path_idx).any(1) | (o.status != plain.status) |
                       (o.layers_done != plain.layers_done)).nonzero().flatten()
                assert bad.numel() == 0, "scenarios %s differ (early_exit %s, cap %s)" % (
                    bad[:8].cpu().numpy(), early_exit, cap)
                assert np.array_equal(o.best_cost.cpu().numpy(), plain.best_cost.cpu().numpy(), equal_nan=True)
                if o.cost_map is not None:
                    assert torch.equal(o.cost_map, plain.cost_map) and torch.equal(o.index_map, plain.index_map)
        assert bool(((plain.status == 0) | (plain.status == 1)).all())
        n_found += int((plain.status == 1).sum())
    print("found %d of %d plans" % (n_found, 2 * B))
