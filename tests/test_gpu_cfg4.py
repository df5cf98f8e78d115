"""The 65536-scenario configuration as its shards see it: the block of scenarios the last of
eight ranks plans (ids 57344..65535 of bench.make_scenarios) through the same calls the
benchmark's step makes, checked by properties over the whole shard, against the oracle on
scenarios from the END of the id range, and through the packed result records."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rank7_shard_of_the_65536_scenario_split(oracle):
    import torch
    import bench
    from motion_planning_amd import batch as mb, dist as mdist
    total, world, rank = bench.CFG4_TOTAL, 8, 7
    lo, hi = mdist.shard_range(total, rank, world)
    assert (lo, hi) == (57344, 65536)
    B, n_s, n_l, n_t = hi - lo, bench.N_S, bench.N_L, bench.ST_N_T
    sc = bench.make_scenarios(lo, B)
    sb = mb.SLBatch(sc["s_layers"], sc["l_row"], sc["ego"], sc["ob_s"], sc["ob_l"], sc["n_ob"], sc["ob_dist"])
    prm = mb.sl_params(early_exit=False, n_l=n_l)
    out = mb.sl_plan(sb, prm, mb.SLPlanOut(B, n_s, n_l, want_maps=False))
    st = mb.st_dp(sc["st_n_s"], sc["down"], sc["up"], sc["n_reg"], sc["cur_vel"], bench.SPEED_LIM, want_tables=False,
                  n_s_max=bench.ST_N_S)
    again = mb.sl_plan(sb, prm, mb.SLPlanOut(B, n_s, n_l, want_maps=False))
    torch.cuda.synchronize()
    status, path = out.status.cpu().numpy(), out.path_idx.cpu().numpy()
    done, best = out.layers_done.cpu().numpy(), out.best_cost.cpu().numpy()
    st_path, st_len = st["path_s_ind"].cpu().numpy(), st["path_len"].cpu().numpy()
    # properties over all 8192 scenarios of the shard
    assert torch.equal(again.path_idx, out.path_idx) and torch.equal(again.status, out.status)   # run to run
    assert set(np.unique(status)) <= {0, 1} and 0 < (status == 1).sum() < B
    assert (done == n_s).all()                                  # every layer relaxed
    found = status == 1
    assert ((path[found] >= 0) & (path[found] < n_l)).all() and (path[~found] == -1).all()
    assert np.isfinite(best[found]).all() and (best[found] <= 1e5).all() and np.isnan(best[~found]).all()
    assert ((st_len >= 2) & (st_len <= n_t)).all() and (st_path[:, 0] == 0).all()
    for b in range(0, B, 97):                                   # s never decreases along an ST plan
        assert (np.diff(st_path[b, :st_len[b]]) >= 0).all() and st_path[b, st_len[b] - 1] < bench.ST_N_S
    # the oracle on scenarios from the end of the id range (ids 65504..65535), found and not found
    tail = list(range(B - 32, B))
    assert found[tail].any() and (~found[tail]).any()

    def reference(b):
        s_map = np.repeat(sc["s_layers"][b][:, None], n_l, axis=1)
        l_map = np.broadcast_to(sc["l_row"][b], (n_s, n_l))
        r = oracle.sl_find_path(s_map, l_map, int((n_l - 1) / 2), sc["ego"][b],
                                np.stack([sc["ob_s"][b], sc["ob_l"][b]], 1), sc["ob_dist"][b], no_early_exit=True)
        s = oracle.st_find_speed(n_t, bench.ST_N_S, sc["down"][b], sc["up"][b], sc["n_reg"][b],
                                 float(sc["cur_vel"][b]), bench.SPEED_LIM)
        return r, s

    with ThreadPoolExecutor(8) as pool:
        refs = list(pool.map(reference, tail))
    for b, (r, s) in zip(tail, refs):
        assert bool(r["found"]) == bool(found[b]) and int(r["layers_done"]) == done[b], b
        if r["found"]:
            assert np.array_equal(path[b], r["path_idx"]), "scenario id %d" % (lo + b)
            want = float(r["cost_map"][-1, r["path_idx"][-1]])
            assert abs(best[b] - want) <= 1e-10 * abs(want)
        assert st_len[b] == len(s["path_s_ind"]) and np.array_equal(st_path[b, :st_len[b]], s["path_s_ind"])
    # the shard's result records: packed on the device, unpacked as the receiving ranks do
    rec = mdist.pack_results(out.path_idx, out.status, out.layers_done, out.best_cost, st["path_s_ind"],
                             st["path_len"])
    assert rec.shape == (B, mdist.record_width(n_s, n_t))
    u = mdist.unpack_results(rec.cpu(), n_s, n_t)
    assert torch.equal(u["path_idx"], out.path_idx.cpu()) and torch.equal(u["status"], out.status.cpu())
    assert torch.equal(u["layers_done"], out.layers_done.cpu()) and torch.equal(u["st_path"], st["path_s_ind"].cpu())
    assert torch.equal(u["st_len"], st["path_len"].cpu())
    assert np.array_equal(u["best_cost"].numpy(), best, equal_nan=True)


def test_all_65536_scenarios_two_independent_gpu_paths_agree(knob):
    """Every scenario of the configuration, not a sample: the fused SL kernel (pruned walk, runs
    of layers without exchange, pair store in LDS) against the materialised path (every edge of
    the lattice priced by the edge kernel into a tensor in HBM, then the plain DP over it), and
    the lanes-per-cell ST DP (cell tables, walks split over four lanes) against the relaxation by
    one thread per cell without tables (MP_ST_GENERIC=2).  Plans must agree bit for bit;
    the oracle pins both paths on samples elsewhere, this pins the rare scenario."""
    import torch
    import bench
    from motion_planning_amd import batch as mb
    total, chunk = bench.CFG4_TOTAL, 2048          # 2048 edge tensors of 64 x 128 x 128 doubles: 17 GB
    n_s, n_l = bench.N_S, bench.N_L
    prm = mb.sl_params(early_exit=False, n_l=n_l)
    n_found = 0
    for lo in range(0, total, chunk):
        sc = bench.make_scenarios(lo, chunk)
        sb = mb.SLBatch(sc["s_layers"], sc["l_row"], sc["ego"], sc["ob_s"], sc["ob_l"], sc["n_ob"], sc["ob_dist"])
        fused = mb.sl_plan(sb, prm, mb.SLPlanOut(chunk, n_s, n_l, want_maps=False))
        edge, est = mb.sl_edge_cost(sb, prm)
        plain = mb.sl_dp(edge, prm)
        del edge
        st_in = (sc["st_n_s"], sc["down"], sc["up"], sc["n_reg"], sc["cur_vel"], bench.SPEED_LIM)
        st = mb.st_dp(*st_in, want_tables=False, n_s_max=bench.ST_N_S)
        knob("MP_ST_GENERIC", 2)
        st_plain = mb.st_dp(*st_in, want_tables=False, n_s_max=bench.ST_N_S)
        knob("MP_ST_GENERIC", -1)
        torch.cuda.synchronize()
        assert bool((est == 1).all())
        bad = ((fused.path_idx != plain.path_idx).any(1) | (fused.status != plain.status)).nonzero().flatten()
        assert bad.numel() == 0, "SL plans differ for scenario ids %s" % (lo + bad[:8].cpu().numpy())
        assert torch.equal(fused.layers_done, plain.layers_done)
        # the instance that writes the maps (every predecessor row in LDS, a smaller pair store): its
        # whole cost map and predecessor map against the plain DP's
        maps = mb.sl_plan(sb, prm, want_maps=True)
        assert torch.equal(maps.cost_map, plain.cost_map) and torch.equal(maps.index_map, plain.index_map)
        assert torch.equal(maps.path_idx, plain.path_idx) and torch.equal(maps.status, plain.status)
        del maps
        assert np.array_equal(fused.best_cost.cpu().numpy(), plain.best_cost.cpu().numpy(), equal_nan=True)
        bad = ((st["path_s_ind"] != st_plain["path_s_ind"]).any(1) | (st["path_len"] != st_plain["path_len"]))
        assert not bool(bad.any()), "ST plans differ for scenario ids %s" % (lo + bad.nonzero().flatten()[:8].cpu().numpy())
        n_found += int((fused.status == 1).sum())
    assert 0 < n_found < total
