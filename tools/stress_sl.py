"""One-off stress of the SL kernels (development aid): the random-batch parity test of
tests/test_gpu_sl.py on fresh seeds and shapes -- fused kernel against the oracle, and against
the edge-tensor path, the shared-row layout and (narrow lattices) the per-node kernel.

    python3 tools/stress_sl.py [cases] [seed]
"""
import sys
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_sl as T
from parity import assert_sl_parity
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    n_l = int(rng.choice([2, 3, 5, 8, 12, 16, 17, 33, 64, 100, 128, 200]))
    n_s = int(rng.integers(2, 14 if n_l > 64 else 24))
    B = int(rng.integers(2, 7))
    gap = float(rng.choice([1.5, 3.0, 8.0, 12.0]))
    n_ob = int(rng.choice([0, 1, 5, 20, 60]))
    r_lo, r_hi = float(rng.choice([0.2, 0.4, 0.8])), float(rng.choice([0.9, 1.5]))
    spread = None if rng.random() < 0.7 else float(rng.uniform(0.5, 3.0))
    jitter = bool(rng.random() < 0.3)
    early = bool(rng.random() < 0.5)
    s_map, l_map, ego, obs, ob_dist = T.random_batch(rng, B, n_s, n_l, gap, n_ob, r_lo, r_hi, spread, jitter)
    mid_l = int((n_l - 1) / 2)
    try:
        res = T.gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early)
        others = [T.gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early, split=True)[0],
                  T.gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early, shared_row=True)]
        if n_l <= 16:
            others.append(T.gpu_sl(s_map, l_map, mid_l, ego, obs, ob_dist, early_exit=early, per_node=True))
        for b in range(B):
            small = n_s * n_l * n_l <= 40000
            ref = oracle.sl_find_path(s_map[b], l_map[b], mid_l, ego[b], obs[b], ob_dist[b],
                                      no_early_exit=not early, want_edges=small)
            assert_sl_parity(res[b], ref, ref.get("edge_cost"))
            for o in (x[b] for x in others):
                assert np.array_equal(res[b]["cost_map"], o["cost_map"])
                assert np.array_equal(res[b]["index_map"], o["index_map"])
                assert np.array_equal(res[b]["path_idx"], o["path_idx"])
                assert res[b]["status"] == o["status"] and res[b]["layers_done"] == o["layers_done"]
        verdict = "ok"
    except AssertionError as e:
        verdict = "MISMATCH %s" % (str(e)[:200],)
        bad += 1
    print("case %2d: B=%d %dx%d gap=%.1f ob=%d r=%.1f..%.1f spread=%s jitter=%s early=%s %s" %
          (c, B, n_s, n_l, gap, n_ob, r_lo, r_hi, spread, jitter, early, verdict), flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
