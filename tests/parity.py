"""Shared parity helpers for the GPU tests."""
import numpy as np

COST_RTOL = 1e-10   # north star asks 1e-5; the fp64 kernels do far better
TIE_RTOL = 1e-9     # index mismatches are allowed only between candidates this close


def assert_costs(got, want, rtol=COST_RTOL):
    np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-12)


def assert_sl_parity(gpu, ref, edge_cost=None):
    """gpu/ref: dicts with found, cost_map, index_map, path_idx, layers_done.

    Costs to COST_RTOL over the relaxed layers; indices exact, except where the
    reference's own competing candidates tie within TIE_RTOL (needs edge_cost)."""
    assert bool(gpu["found"]) == bool(ref["found"])
    d = int(ref["layers_done"])
    assert int(gpu["layers_done"]) == d
    assert_costs(gpu["cost_map"][:d], ref["cost_map"][:d])
    gi, ri = np.asarray(gpu["index_map"][:d]), np.asarray(ref["index_map"][:d])
    bad = np.argwhere(gi != ri)
    for s, l in bad:
        assert edge_cost is not None, "index mismatch at layer %d lane %d" % (s, l)
        cand = edge_cost[s, l, :] + ref["cost_map"][s - 1]
        a, b = cand[gi[s, l]], cand[ri[s, l]]
        assert abs(a - b) <= TIE_RTOL * max(abs(a), abs(b)), "index mismatch beyond a tie at (%d,%d)" % (s, l)
    if ref["found"] and len(bad) == 0:
        assert np.array_equal(gpu["path_idx"], ref["path_idx"])


def pad_obstacles(obs, stride=None):
    """list of [n_i, 2] arrays -> (ob_s [B, stride], ob_l, n_ob)."""
    B = len(obs)
    n = np.array([len(o) for o in obs], dtype=np.int32)
    stride = stride or max(1, int(n.max()))
    ob_s = np.zeros((B, stride))
    ob_l = np.zeros((B, stride))
    for b, o in enumerate(obs):
        o = np.asarray(o, dtype=np.float64).reshape(-1, 2)
        ob_s[b, :len(o)] = o[:, 0]
        ob_l[b, :len(o)] = o[:, 1]
    return ob_s, ob_l, n
