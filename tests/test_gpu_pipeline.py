"""The batched full pipeline (reference-line tables -> C->F -> SL DP -> path
sampling -> ST intervals -> ST DP -> speed profile -> F->C) on the GPU against
the reference's recorded pipeline scenes and the CPU oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-11):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), b, rtol=rtol, atol=1e-12)


def pack_scenes(scenes):
    """scenes: list of dict(rx, ry, ego_vel, st_pos [n,2], st_rad [n], dy_pos [m,11,2], dy_rad [m])."""
    B = len(scenes)
    nk = max(len(s["rx"]) for s in scenes)
    Ns = max(1, max(len(s["st_rad"]) for s in scenes))
    Nd = max(1, max(len(s["dy_rad"]) for s in scenes))
    kx, ky = np.zeros((B, nk)), np.zeros((B, nk))
    n_knots = np.zeros(B, np.int32)
    ego_vel = np.zeros((B, 2))
    st_pos, st_rad, n_st = np.zeros((B, Ns, 2)), np.zeros((B, Ns)), np.zeros(B, np.int32)
    dy_pos, dy_rad, n_dy = np.zeros((B, Nd, 11, 2)), np.zeros((B, Nd)), np.zeros(B, np.int32)
    for b, s in enumerate(scenes):
        n = len(s["rx"])
        n_knots[b] = n
        kx[b, :n], ky[b, :n] = s["rx"], s["ry"]
        kx[b, n:] = s["rx"][-1]
        ego_vel[b] = s["ego_vel"]
        n_st[b], n_dy[b] = len(s["st_rad"]), len(s["dy_rad"])
        st_pos[b, :n_st[b]] = np.asarray(s["st_pos"]).reshape(-1, 2)
        st_rad[b, :n_st[b]] = s["st_rad"]
        dy_pos[b, :n_dy[b]] = np.asarray(s["dy_pos"]).reshape(-1, 11, 2)
        dy_rad[b, :n_dy[b]] = s["dy_rad"]
    return dict(knots_x=kx, knots_y=ky, n_knots=n_knots, ego_vel=ego_vel, st_pos=st_pos, st_rad=st_rad, n_st=n_st,
                dy_pos=dy_pos, dy_rad=dy_rad, n_dy=n_dy)


def test_pipeline_golden_scenes_in_one_batch():
    import torch
    from motion_planning_amd import pipeline as pl
    g = load_golden("pipe.npz")
    cases = [str(c) for c in g["cases"]]
    scenes = [dict(rx=g[c + "_rx"], ry=g[c + "_ry"], ego_vel=g[c + "_ego_vel"], st_pos=g[c + "_st_pos"],
                   st_rad=g[c + "_st_rad"], dy_pos=g[c + "_dy_pos"], dy_rad=g[c + "_dy_rad"]) for c in cases]
    r = pl.plan_batch(**pack_scenes(scenes), want_maps=True)
    torch.cuda.synchronize()
    status = r["status"].cpu().numpy()
    for b, c in enumerate(cases):
        p = c + "_"
        assert (status[b] == pl.PIPE_OK) == bool(g[p + "found"]), c
        n_s = int(r["n_s"][b])
        assert n_s == g[p + "s_map"].shape[0]
        close(r["s_layers"][b, :n_s].cpu().numpy(), g[p + "s_map"][:, 0])
        close(r["ego_sl"][b].cpu().numpy(), g[p + "ego"])
        close(r["robot_vel"][b].cpu().numpy(), g[p + "robot_vel"])
        n_ob = int(r["n_ob"][b])
        assert n_ob == len(g[p + "ob"])
        close(np.stack([r["ob_s"][b, :n_ob].cpu().numpy(), r["ob_l"][b, :n_ob].cpu().numpy()], 1), g[p + "ob"])
        assert float(r["ob_dist"][b]) == float(g[p + "ob_dist"])
        done = int(np.sum(np.any(g[p + "index_map"] != -2, axis=1)))
        close(r["cost_map"][b][:done].cpu().numpy(), g[p + "cost_map"][:done], rtol=1e-10)
        assert np.array_equal(r["index_map"][b][:done].cpu().numpy(), g[p + "index_map"][:done])
        if status[b] != pl.PIPE_OK:
            assert status[b] == pl.PIPE_NOT_FOUND
            continue
        assert np.array_equal(r["path_idx"][b, :n_s].cpu().numpy(), g[p + "path_idx"])
        k = b   # every result array is indexed by scene
        P = int(r["n_path"][k])
        assert P == len(g[p + "path_sl"])
        close(np.stack([r["path_s"][k, :P].cpu().numpy(), r["path_l"][k, :P].cpu().numpy()], 1), g[p + "path_sl"])
        assert int(r["st_n_s"][k]) == int(g[p + "st_n_s"])
        want_n = g[p + "st_n_reg"]
        assert np.array_equal(r["st_n_reg"][k].cpu().numpy(), want_n)
        for i in range(11):
            close(r["st_down"][k, i, :want_n[i]].cpu().numpy(), g[p + "st_down"][i, :want_n[i]])
            close(r["st_up"][k, i, :want_n[i]].cpu().numpy(), g[p + "st_up"][i, :want_n[i]])
        ns = int(g[p + "st_n_s"])
        close(r["st_cost"][k, :, :ns].cpu().numpy(), g[p + "st_cost"], rtol=1e-10)
        assert np.array_equal(r["st_pre"][k, :, :ns].cpu().numpy(), g[p + "st_pre"])
        plen = int(r["st_path_len"][k])
        assert np.array_equal(r["st_path_s_ind"][k, :plen].cpu().numpy(), g[p + "path_s_ind"])
        n = int(r["n_prof"][k])
        assert n == len(g[p + "ss"])
        close(r["ss"][k, :n].cpu().numpy(), g[p + "ss"])
        close(r["vv"][k, :n].cpu().numpy(), g[p + "vv"])
        close(np.stack([r["final_s"][k, :n].cpu().numpy(), r["final_l"][k, :n].cpu().numpy()], 1), g[p + "path_sl2"])
        close(np.stack([r["final_x"][k, :n].cpu().numpy(), r["final_y"][k, :n].cpu().numpy()], 1), g[p + "path_xy"])


def random_scene(rng, n_st, n_dy):
    nk = int(rng.integers(13, 22))
    xs = -2.0 + 4.5 * np.arange(nk)
    c = rng.uniform(-0.006, 0.006)
    ys = c * xs * xs
    st_x = rng.uniform(0, xs[-1], n_st)
    st_pos = np.stack([st_x, c * st_x * st_x + rng.uniform(-6, 6, n_st)], 1)
    dy0 = np.stack([rng.uniform(0, xs[-1], n_dy), rng.uniform(-6, 6, n_dy)], 1)
    vel = np.stack([rng.uniform(-3, 3, n_dy), rng.uniform(-0.5, 0.5, n_dy)], 1)
    frames = [dy0]
    for _ in range(10):
        frames.append(frames[-1] + vel * 0.5)
    return dict(rx=xs, ry=ys, ego_vel=np.array([rng.uniform(3, 8), rng.uniform(-0.5, 0.5)]), st_pos=st_pos,
                st_rad=np.full(n_st, 0.8), dy_pos=np.stack(frames, 1), dy_rad=np.full(n_dy, 0.8))


@pytest.mark.parametrize("n_l", [5, 12, 128])
def test_pipeline_random_vs_oracle(oracle, n_l):
    """Curved 13-21 knot lines, a mix of sparse and crowded scenes; the default 5 lanes
    (lane-per-edge SL kernel) and wider lattices (12: the same kernel at 16 lanes per target, 128: the
    workgroup-per-scene SL kernel inside the chain)."""
    import torch
    from motion_planning_amd import pipeline as pl
    rng = np.random.default_rng(17 + n_l)
    scenes = [random_scene(rng, *cnt) for cnt in [(6, 2), (12, 3), (24, 6), (3, 0), (0, 0), (0, 2), (40, 8), (10, 4),
                                                  (192, 64), (5, 1)]]
    r = pl.plan_batch(**pack_scenes(scenes), n_l=n_l)
    torch.cuda.synchronize()
    status = r["status"].cpu().numpy()
    n_found = 0
    for b, sc in enumerate(scenes):
        st = [(pos, rad) for pos, rad in zip(sc["st_pos"], sc["st_rad"])]
        dy = [(fr, rad) for fr, rad in zip(sc["dy_pos"], sc["dy_rad"])]
        try:
            sl = oracle.SLMap(sc["rx"], sc["ry"], 0.0, sc["ego_vel"], st, dy, n_l=n_l)
        except ValueError:   # Frenet map longer than 100 m
            assert status[b] == pl.PIPE_MAP_TOO_BIG
            continue
        assert int(r["n_s"][b]) == sl.n_s
        close(r["s_layers"][b, :sl.n_s].cpu().numpy(), sl.s_map[:, 0])
        n_ob = int(r["n_ob"][b])
        assert n_ob == len(sl.ob_list)
        if n_ob:
            close(np.stack([r["ob_s"][b, :n_ob].cpu().numpy(), r["ob_l"][b, :n_ob].cpu().numpy()], 1),
                  np.array(sl.ob_list))
        found, curve, ref = oracle.sl_plan(sl)
        assert (status[b] == pl.PIPE_OK) == found, b
        if not found:
            continue
        n_found += 1
        assert np.array_equal(r["path_idx"][b, :sl.n_s].cpu().numpy(), ref["path_idx"])
        k = b   # every result array is indexed by scene
        pb = sl.path_sampling(curve)
        P = int(r["n_path"][k])
        assert P == len(pb)
        close(np.stack([r["path_s"][k, :P].cpu().numpy(), r["path_l"][k, :P].cpu().numpy()], 1), pb)
        n_s, n_t = oracle.st_dims(pb[-1][0])
        assert int(r["st_n_s"][k]) == n_s
        down, up, n_reg = oracle.st_add_obstacle(n_t, pb, sl.st_ob_traj, sl.dy_ob_traj)
        assert np.array_equal(r["st_n_reg"][k].cpu().numpy(), n_reg)
        sr = oracle.st_find_speed(n_t, n_s, down, up, n_reg, sl.robot_vel, 25.0 * 5.0 / 18.0)
        plen = int(r["st_path_len"][k])
        assert np.array_equal(r["st_path_s_ind"][k, :plen].cpu().numpy(), sr["path_s_ind"])
        ss, vv = oracle.speed_profile(sr["path_s_ind"], sl.robot_vel)
        n = int(r["n_prof"][k])
        assert n == len(ss)
        close(r["ss"][k, :n].cpu().numpy(), ss)
        close(r["vv"][k, :n].cpu().numpy(), vv, rtol=1e-10)
        pb2 = sl.path_sampling(curve, ss)
        close(np.stack([r["final_s"][k, :n].cpu().numpy(), r["final_l"][k, :n].cpu().numpy()], 1), pb2)
        close(np.stack([r["final_x"][k, :n].cpu().numpy(), r["final_y"][k, :n].cpu().numpy()], 1),
              sl.path_convert(pb2), rtol=1e-10)
    assert 0 < n_found
    print("pipeline: %d of %d scenes found" % (n_found, len(scenes)))


@pytest.mark.parametrize("line", ["straight", "curved", "steep"])
def test_front_end_projection_equals_the_sample_walk(oracle, line):
    """The chain's front end finds an obstacle's nearest reference-line sample through
    bounding boxes of sample blocks; mp_c2f_batch walks the samples one by one.  Same
    tables, same points: the same bits, also where two samples are equally far (points
    above the middle between two samples of a straight line), for points far from the
    line, before its start and behind its end."""
    import torch
    from motion_planning_amd import batch as mb
    from motion_planning_amd import pipeline as pl
    rng = np.random.default_rng({"straight": 1, "curved": 2, "steep": 3}[line])
    c = {"straight": 0.0, "curved": 0.004, "steep": -0.011}[line]
    scenes = []
    for b in range(6):
        nk = 14 + b
        xs = -2.0 + 4.5 * np.arange(nk)
        n_st, n_dy = 300, 20
        st_x = rng.uniform(-12.0, xs[-1] + 12.0, n_st)
        st_y = c * st_x * st_x + rng.uniform(-45.0, 45.0, n_st) * rng.choice([1.0, 0.2, 0.01], n_st)
        # exactly above / below the middle between two samples, and on samples
        st_x[:120] = -2.0 + 0.25 * rng.integers(0, 4 * int(xs[-1]), 120)
        if line == "straight":
            st_y[:40] = rng.choice([-3.0, -0.5, 0.0, 0.75, 6.0], 40)
        dy0 = np.stack([rng.uniform(0, xs[-1], n_dy), rng.uniform(-20, 20, n_dy)], 1)
        vel = np.stack([rng.uniform(-6, 6, n_dy), rng.uniform(-2, 2, n_dy)], 1)
        frames = [dy0]
        for _ in range(10):
            frames.append(frames[-1] + vel * 0.5)
        scenes.append(dict(rx=xs, ry=c * xs * xs, ego_vel=np.array([5.0, 0.0]), st_pos=np.stack([st_x, st_y], 1),
                           st_rad=np.full(n_st, 0.8), dy_pos=np.stack(frames, 1), dy_rad=np.full(n_dy, 0.8)))
    packed = pack_scenes(scenes)
    r = pl.plan_batch(**packed)
    torch.cuda.synchronize()
    ok = (r["rl_status"].cpu().numpy() == 1) & (r["status"].cpu().numpy() != pl.PIPE_MAP_TOO_BIG)
    assert ok.sum() >= 3
    B = len(scenes)
    n_st, n_dy = packed["st_pos"].shape[1], packed["dy_pos"].shape[1]
    scn_st = np.repeat(np.arange(B, dtype=np.int32), n_st)
    s, l = mb.c2f(r["tables"], packed["st_pos"][..., 0], packed["st_pos"][..., 1], scn_st)
    scn_dy = np.repeat(np.arange(B, dtype=np.int32), n_dy * 11)
    sd, ld = mb.c2f(r["tables"], packed["dy_pos"][..., 0], packed["dy_pos"][..., 1], scn_dy)
    torch.cuda.synchronize()
    for b in np.nonzero(ok)[0]:
        assert torch.equal(r["st_s"][b, :n_st], s.reshape(B, n_st)[b])
        assert torch.equal(r["st_l"][b, :n_st], l.reshape(B, n_st)[b])
        assert torch.equal(r["dy_s"][b].reshape(-1)[:n_dy * 11], sd.reshape(B, -1)[b])
        assert torch.equal(r["dy_l"][b].reshape(-1)[:n_dy * 11], ld.reshape(B, -1)[b])
    # and the oracle's conversion of a sample of the points of one scene
    b = int(np.nonzero(ok)[0][0])
    sc = scenes[b]
    conv = oracle.CartesianFrenetConverter(0.0, 0.0, sc["rx"], sc["ry"], 0.0)
    for k in list(range(0, 120, 7)) + list(range(120, 300, 11)):
        ws, wl = conv.cartesian_to_frenet(sc["st_pos"][k, 0], sc["st_pos"][k, 1])
        close([float(r["st_s"][b, k]), float(r["st_l"][b, k])], [ws, wl])


def test_pipeline_map_too_big_is_a_status():
    import torch
    from motion_planning_amd import pipeline as pl
    xs = -2.0 + 4.5 * np.arange(30)   # 128 m reference line
    ok = dict(rx=-2.0 + 4.5 * np.arange(10), ry=np.zeros(10), ego_vel=np.array([5.0, 0.0]), st_pos=np.zeros((0, 2)),
              st_rad=np.zeros(0), dy_pos=np.zeros((0, 11, 2)), dy_rad=np.zeros(0))
    big = dict(ok, rx=xs, ry=np.zeros(30))
    r = pl.plan_batch(**pack_scenes([ok, big]))
    torch.cuda.synchronize()
    assert list(r["status"].cpu().numpy()) == [pl.PIPE_OK, pl.PIPE_MAP_TOO_BIG]
    # knots already on the device: no read back to size the tables, which then hold the samples
    # of a 100 m map; a line with more samples than that (here 128 m, and one of 300 m) is the
    # same "map too big", not a table error
    from motion_planning_amd import batch as mb
    huge = dict(ok, rx=-2.0 + 10.0 * np.arange(31), ry=np.zeros(31))
    packed = pack_scenes([ok, big, huge])
    for k in ("knots_x", "knots_y"):
        packed[k] = mb.to_dev(packed[k])
    r = pl.plan_batch(**packed)
    torch.cuda.synchronize()
    assert r["dims"].m_max == pl.M_MAX_MAP
    assert list(r["status"].cpu().numpy()) == [pl.PIPE_OK, pl.PIPE_MAP_TOO_BIG, pl.PIPE_MAP_TOO_BIG]


def test_unsampled_result_is_not_reported_as_planned():
    """A scene whose speed profile or final path points could not be produced (here: fewer
    result slots than the profile needs) must not keep status PIPE_OK with zero-filled path
    points: the last kernel folds the outcome of both spline evaluations into `status`.
    With enough slots the same scene plans."""
    import torch
    from motion_planning_amd import batch as mb, pipeline as pl
    sc = dict(rx=-2.0 + 4.5 * np.arange(10), ry=np.zeros(10), ego_vel=np.array([5.0, 0.0]),
              st_pos=np.zeros((0, 2)), st_rad=np.zeros(0), dy_pos=np.zeros((0, 11, 2)), dy_rad=np.zeros(0))
    p = pack_scenes([sc, sc])
    B = 2
    dev = dict(knots_x=mb.to_dev(p["knots_x"]), knots_y=mb.to_dev(p["knots_y"]),
               n_knots=mb.to_dev(p["n_knots"], torch.int32), ego_vel=mb.to_dev(p["ego_vel"]),
               st_pos=mb.to_dev(p["st_pos"]), st_rad=mb.to_dev(p["st_rad"]), n_st=mb.to_dev(p["n_st"], torch.int32),
               dy_pos=mb.to_dev(p["dy_pos"]), dy_rad=mb.to_dev(p["dy_rad"]), n_dy=mb.to_dev(p["n_dy"], torch.int32),
               line_vec=torch.zeros(B, dtype=torch.float64, device="cuda"))
    params = pl.default_params()
    dims = pl.plan_dims(B, p["knots_x"].shape[1], 1, 1, m_max=232, params=params)
    a = pl.plan_scenes(dev, dims, params).arrays
    torch.cuda.synchronize()
    assert list(a["status"].cpu().numpy()) == [pl.PIPE_OK] * 2
    n = int(a["n_prof"][0])
    assert n > 20 and int(a["n_final"][0]) == n
    dims.q_max = 20   # the profile has n > 20 samples
    a = pl.plan_scenes(dev, dims, params).arrays
    torch.cuda.synchronize()
    assert list(a["status"].cpu().numpy()) == [pl.PIPE_ST_CAPACITY] * 2
    assert int(a["n_final"][0]) == 0


def test_bad_scenes_do_not_disturb_their_neighbours():
    """Scenes with NaN / infinite coordinates, negative or oversized counts and a
    one-knot line share a batch with healthy scenes: the call returns, the healthy
    scenes come out bit for bit as in a clean batch, and no damaged scene reports a
    plan built on NaNs as usable geometry (its status is an error code or its
    outputs are absent)."""
    import torch
    from motion_planning_amd import pipeline as pl
    rng = np.random.default_rng(91)
    scenes = [random_scene(rng, 20, 4) for _ in range(10)]
    clean = pack_scenes(scenes)
    bad = {k: v.copy() for k, v in clean.items()}
    bad["st_pos"][2, 3, 0] = np.nan            # an obstacle at NaN
    bad["dy_pos"][3, 1, 4, 1] = np.inf         # one predicted frame at infinity
    bad["knots_y"][4, 5] = np.inf              # a reference-line knot at infinity
    bad["knots_x"][5, 2] = np.nan              # a NaN knot abscissa
    bad["ego_vel"][6] = np.nan                 # no ego velocity
    bad["n_st"][7], bad["n_dy"][7] = -3, 1000  # counts outside the arrays
    bad["n_knots"][8] = 1                      # a line of one knot
    good = [0, 1, 9]
    r0 = pl.plan_batch(**clean, m_max=232)
    r1 = pl.plan_batch(**bad, m_max=232)
    torch.cuda.synchronize()
    for k in ("status", "n_layers", "path_idx", "n_ob", "st_n_reg", "st_path_s_ind", "n_prof", "ss", "vv",
              "final_x", "final_y"):
        a, b = r0[k][good].cpu().numpy(), r1[k][good].cpu().numpy()
        assert np.array_equal(a, b, equal_nan=True), k
    st = r1["status"].cpu().numpy()
    assert st[8] == pl.PIPE_BAD_LINE
    # the counts of scene 7 were clamped to the arrays: it plans like a scene without statics
    assert st[7] in (pl.PIPE_OK, pl.PIPE_NOT_FOUND)
    n_prof = r1["n_prof"].cpu().numpy()
    fx = r1["final_x"].cpu().numpy()
    for b in (4, 5):    # a broken reference line never yields a path
        assert st[b] != pl.PIPE_OK or not np.isfinite(fx[b, :n_prof[b]]).all() or n_prof[b] == 0, (b, st[b])


def test_plan_scenes_replays_as_a_hip_graph():
    """mp_plan_scenes only enqueues work on the caller's stream (no allocation, no
    synchronisation, nothing read back), so a caller can capture it in a HIP graph and
    replay it on new inputs written into the same buffers."""
    import torch
    from motion_planning_amd import batch as mb, pipeline as pl
    rng = np.random.default_rng(4)
    packs = [pack_scenes([random_scene(rng, 12, 3) for _ in range(5)]) for _ in range(2)]
    nk = max(p["knots_x"].shape[1] for p in packs)

    def dev_scenes(p):
        B = p["knots_x"].shape[0]
        kx, ky = np.zeros((B, nk)), np.zeros((B, nk))
        kx[:, :p["knots_x"].shape[1]], ky[:, :p["knots_y"].shape[1]] = p["knots_x"], p["knots_y"]
        d = dict(knots_x=mb.to_dev(kx), knots_y=mb.to_dev(ky), n_knots=mb.to_dev(p["n_knots"], torch.int32),
                 ego_vel=mb.to_dev(p["ego_vel"]), st_pos=mb.to_dev(p["st_pos"]), st_rad=mb.to_dev(p["st_rad"]),
                 n_st=mb.to_dev(p["n_st"], torch.int32), dy_pos=mb.to_dev(p["dy_pos"]),
                 dy_rad=mb.to_dev(p["dy_rad"]), n_dy=mb.to_dev(p["n_dy"], torch.int32),
                 line_vec=torch.zeros(B, dtype=torch.float64, device="cuda"))
        return d

    params = pl.default_params()
    dims = pl.plan_dims(5, nk, 12, 3, m_max=232, params=params)
    eager = []
    for p in packs:
        a = pl.plan_scenes(dev_scenes(p), dims, params)
        torch.cuda.synchronize()
        eager.append({k: v.clone() for k, v in a.arrays.items() if k in ("status", "path_idx", "vv", "final_x", "final_y")})
    buf = dev_scenes(packs[0])
    arena = pl.PlanArena(dims)
    pl.plan_scenes(buf, dims, params, arena)        # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pl.plan_scenes(buf, dims, params, arena)
    for p, want in zip(packs, eager):
        for k, v in dev_scenes(p).items():
            buf[k].copy_(v)
        g.replay()
        torch.cuda.synchronize()
        for k, v in want.items():
            assert torch.equal(arena.arrays[k], v), k


def test_pipeline_full_size_batch_properties_and_sample_vs_oracle(oracle):
    """The benchmark's whole-replan workload (4096 scenes, 192 static + 64 moving obstacles
    each, curved lines): two runs agree bit for bit, every status is one the reference
    can produce, per-scene invariants hold for all scenes, and a sample of scenes of every
    outcome matches the oracle's chain end to end."""
    import torch
    import bench
    from motion_planning_amd import batch as mb, pipeline as pl
    B = 4096
    sc = bench.make_scenes(B)
    dev = {k: mb.to_dev(v, torch.int32 if v.dtype == np.int32 else torch.float64) for k, v in sc.items()}
    dev["line_vec"] = torch.zeros(B, dtype=torch.float64, device="cuda")
    params = pl.default_params()
    dims = pl.plan_dims(B, sc["knots_x"].shape[1], 192, 64, n_l=5, m_max=200, params=params)
    a1 = pl.plan_scenes(dev, dims, params)
    a2 = pl.plan_scenes(dev, dims, params)
    torch.cuda.synchronize()
    r = a1.arrays
    for k in ("status", "path_idx", "n_ob", "ob_s", "st_n_reg", "st_path_s_ind", "ss", "vv", "final_x", "final_y"):
        assert torch.equal(r[k], a2.arrays[k]), k
    status = r["status"].cpu().numpy()
    assert set(np.unique(status)) <= {pl.PIPE_OK, pl.PIPE_NOT_FOUND, pl.PIPE_MAP_TOO_BIG}
    assert (status == pl.PIPE_OK).mean() > 0.3 and (status == pl.PIPE_NOT_FOUND).any()
    n_layers = r["n_layers"].cpu().numpy()
    path_idx = r["path_idx"].cpu().numpy()
    n_prof, plen = r["n_prof"].cpu().numpy(), r["st_path_len"].cpu().numpy()
    st_idx = r["st_path_s_ind"].cpu().numpy()
    vv, ss = r["vv"].cpu().numpy(), r["ss"].cpu().numpy()
    ok = status == pl.PIPE_OK
    assert (n_layers[status == pl.PIPE_MAP_TOO_BIG] == 0).all() and (n_layers[ok] >= 2).all()
    assert (n_prof[~ok] == 0).all() and (n_prof[ok] == (plen[ok] - 1) * 5 + 1).all()
    for b in np.flatnonzero(ok):
        n = n_layers[b]
        assert (path_idx[b, :n] >= 0).all() and (path_idx[b, :n] < 5).all() and (path_idx[b, n:] == -1).all()
        p = st_idx[b, :plen[b]]
        assert p[0] == 0 and (np.diff(p) >= 0).all()              # s never decreases along the speed plan
    assert np.allclose(vv[ok, 0], r["robot_vel"].cpu().numpy()[ok])  # the profile starts at the ego's speed
    assert (ss[ok, 0] == 0.0).all()
    sample = (list(np.flatnonzero(ok)[:10]) + list(np.flatnonzero(status == pl.PIPE_NOT_FOUND)[:5]) +
              list(np.flatnonzero(status == pl.PIPE_MAP_TOO_BIG)[:3]))
    for b in sample:
        nk = int(sc["n_knots"][b])
        st = [(p, rad) for p, rad in zip(sc["st_pos"][b], sc["st_rad"][b])]
        dy = [(fr, rad) for fr, rad in zip(sc["dy_pos"][b], sc["dy_rad"][b])]
        try:
            sl = oracle.SLMap(sc["knots_x"][b, :nk], sc["knots_y"][b, :nk], 0.0, sc["ego_vel"][b], st, dy)
        except ValueError:
            assert status[b] == pl.PIPE_MAP_TOO_BIG
            continue
        assert n_layers[b] == sl.n_s and int(r["n_ob"][b]) == len(sl.ob_list)
        found, curve, ref = oracle.sl_plan(sl)
        assert (status[b] == pl.PIPE_OK) == found
        if not found:
            continue
        assert np.array_equal(path_idx[b, :sl.n_s], ref["path_idx"])
        pb = sl.path_sampling(curve)
        n_s, n_t = oracle.st_dims(pb[-1][0])
        down, up, n_reg = oracle.st_add_obstacle(n_t, pb, sl.st_ob_traj, sl.dy_ob_traj)
        assert np.array_equal(r["st_n_reg"][b].cpu().numpy(), n_reg)
        sr = oracle.st_find_speed(n_t, n_s, down, up, n_reg, sl.robot_vel, 25.0 * 5.0 / 18.0)
        assert np.array_equal(st_idx[b, :plen[b]], sr["path_s_ind"])
        want_ss, want_vv = oracle.speed_profile(sr["path_s_ind"], sl.robot_vel)
        n = int(n_prof[b])
        assert n == len(want_ss)
        close(vv[b, :n], want_vv, rtol=1e-10)
        close(np.stack([r["final_x"][b, :n].cpu().numpy(), r["final_y"][b, :n].cpu().numpy()], 1),
              sl.path_convert(sl.path_sampling(curve, want_ss)), rtol=1e-10)


def test_two_parts_on_two_streams_equal_the_single_call():
    """The schedule the benchmark's headline uses: the scenes cut into two parts, each
    planned by its own mp_plan_scenes call on its own stream with its own arena, several
    steps in a row with no part waiting for the other -- same results, bit for bit, as one
    call over all scenes."""
    import torch
    import bench
    from motion_planning_amd import batch as mb, pipeline as pl
    B = 1001                                        # parts of 501 and 500 scenes
    sc = bench.make_scenes(B, n_st=40, n_dy=12, seed=11)
    dev = {k: mb.to_dev(v, torch.int32 if v.dtype == np.int32 else torch.float64) for k, v in sc.items()}
    dev["line_vec"] = torch.zeros(B, dtype=torch.float64, device="cuda")
    params = pl.default_params()
    whole = pl.plan_scenes(dev, pl.plan_dims(B, sc["knots_x"].shape[1], 40, 12, n_l=5, m_max=200, params=params),
                           params)
    parts = []
    for lo, hi in ((0, 501), (501, B)):
        dims = pl.plan_dims(hi - lo, sc["knots_x"].shape[1], 40, 12, n_l=5, m_max=200, params=params)
        parts.append(dict(dev={k: v[lo:hi].contiguous() for k, v in dev.items()}, dims=dims, arena=pl.PlanArena(dims),
                          stream=torch.cuda.Stream()))
    torch.cuda.synchronize()
    for _ in range(3):
        for p in parts:
            with torch.cuda.stream(p["stream"]):
                pl.plan_scenes(p["dev"], p["dims"], params, p["arena"])
    torch.cuda.synchronize()
    status = whole.arrays["status"]
    assert (status == pl.PIPE_OK).any() and (status != pl.PIPE_OK).any()
    for k in ("status", "sl_status", "path_idx", "n_ob", "st_n_s", "st_n_reg", "st_path_s_ind", "st_path_len",
              "n_prof", "n_final", "ss", "vv", "final_x", "final_y"):
        got = torch.cat([p["arena"].arrays[k] for p in parts])
        want = whole.arrays[k]
        assert got.shape == want.shape, k
        if got.is_floating_point():
            assert torch.equal(got.isnan(), want.isnan()), k
            got, want = got.nan_to_num(), want.nan_to_num()
        assert torch.equal(got, want), k


def test_full_size_batch_every_stage_against_an_independent_kernel(knob):
    """All 4096 scenes of the benchmark's whole-replan workload, not a sample: what four stages
    of the one-call chain left in the arena against kernels that reach the same results another
    way, from the chain's own intermediates.  Obstacle frames to Frenet coordinates: the front
    end's search through bounding boxes of sample blocks against mp_c2f_batch's walk over the
    samples.  Path search: the lane-per-edge kernel against the fused kernel (thread per node,
    pruned walk).  Blocked intervals: every scan once with a work list against a scan per time
    layer.  Speed search: lanes per cell with cell tables against one thread per cell.  Bit for
    bit; the oracle pins each of them on samples elsewhere."""
    import types
    import torch
    import bench
    from motion_planning_amd import batch as mb, pipeline as pl
    B = 4096
    sc = bench.make_scenes(B)
    dev = {k: mb.to_dev(v, torch.int32 if v.dtype == np.int32 else torch.float64) for k, v in sc.items()}
    dev["line_vec"] = torch.zeros(B, dtype=torch.float64, device="cuda")
    params = pl.default_params()
    dims = pl.plan_dims(B, sc["knots_x"].shape[1], 192, 64, n_l=5, m_max=200, params=params)
    r = pl.plan_scenes(dev, dims, params).arrays
    torch.cuda.synchronize()
    status = r["status"]
    live = (status == pl.PIPE_OK) | (status == pl.PIPE_NOT_FOUND)       # scenes the front end took
    assert int(live.sum()) > B // 2
    # 1. obstacle frames in Frenet coordinates
    tabs = types.SimpleNamespace(m_max=dims.m_max, tab_x=r["tab_x"], tab_y=r["tab_y"], tab_s=r["tab_s"],
                                 tab_theta=r["tab_theta"], tab_kappa=r["tab_kappa"], M=r["tab_m"])
    scn = torch.arange(B, dtype=torch.int32, device="cuda")
    s, l = mb.c2f(tabs, dev["st_pos"][..., 0], dev["st_pos"][..., 1], scn[:, None].expand(B, 192).contiguous())
    assert torch.equal(s.reshape(B, 192)[live], r["st_s"][live]) and torch.equal(l.reshape(B, 192)[live], r["st_l"][live])
    F = dims.n_frames
    s, l = mb.c2f(tabs, dev["dy_pos"][..., 0], dev["dy_pos"][..., 1], scn[:, None, None].expand(B, 64, F).contiguous())
    assert torch.equal(s.reshape(B, 64, F)[live], r["dy_s"][live]) and torch.equal(l.reshape(B, 64, F)[live], r["dy_l"][live])
    # 2. path search on the chain's lattice and obstacle list
    n_l = dims.n_l
    sb = mb.SLBatch(r["s_layers"], r["l_row"][None].expand(B, n_l).contiguous(), r["ego_sl"], r["ob_s"], r["ob_l"],
                    r["n_ob"], r["ob_dist"], n_layers=r["n_layers"], check=False)
    knob("MP_SL_NARROW", 0)
    fused = mb.sl_plan(sb, mb.sl_params(early_exit=True, n_l=n_l), want_maps=False)
    knob("MP_SL_NARROW", -1)
    torch.cuda.synchronize()
    assert torch.equal(fused.status[live], r["sl_status"][live])
    assert torch.equal(fused.path_idx[live], r["path_idx"][live])
    assert torch.equal(fused.layers_done[live], r["layers_done"][live])
    assert np.array_equal(fused.best_cost[live].cpu().numpy(), r["best_cost"][live].cpu().numpy(), equal_nan=True)
    # 3. blocked intervals from the chain's sampled path
    ok = status == pl.PIPE_OK
    n_t = dims.n_t
    knob("MP_ST_REGIONS_ONCE", 0)
    down, up, n_reg, reg_st = mb.st_regions(n_t, r["path_s"], r["path_l"], r["n_path"], r["st_s"], r["st_l"],
                                            dev["st_rad"], dev["n_st"], r["dy_s"][:, :, :n_t].contiguous(),
                                            r["dy_l"][:, :, :n_t].contiguous(), dev["dy_rad"], dev["n_dy"], dims.r_max)
    knob("MP_ST_REGIONS_ONCE", -1)
    torch.cuda.synchronize()
    assert torch.equal(n_reg[ok], r["st_n_reg"][ok]) and torch.equal(reg_st[ok], r["reg_status"][ok])
    used = torch.arange(dims.r_max, device="cuda")[None, None, :] < n_reg[:, :, None]
    assert torch.equal((down * used)[ok], (r["st_down"] * used)[ok]) and torch.equal((up * used)[ok], (r["st_up"] * used)[ok])
    # 4. speed search on the chain's intervals
    knob("MP_ST_GENERIC", 2)
    st = mb.st_dp(r["st_n_s"], r["st_down"], r["st_up"], r["st_n_reg"], r["robot_vel"], params.speed_lim,
                  ds=params.st_ds, dt=params.st_dt, want_tables=False, n_s_max=dims.st_n_s_max)
    knob("MP_ST_GENERIC", -1)
    torch.cuda.synchronize()
    assert torch.equal(st["path_len"][ok], r["st_path_len"][ok])
    assert torch.equal(st["path_s_ind"][ok], r["st_path_s_ind"][ok])
    # 5. the stages that size their workgroup by the batch (one wave per scene in a large batch):
    # the same bits with the workgroups a small batch gets
    keep = {k: r[k].clone() for k in ("status", "tab_s", "tab_theta", "path_s", "path_l", "n_path", "st_n_s", "ss", "vv",
                                      "final_x", "final_y", "n_prof")}
    knob("MP_RL_THREADS", 256)
    knob("MP_CV_THREADS", 128)
    r2 = pl.plan_scenes(dev, dims, params).arrays
    knob("MP_RL_THREADS", -1)
    knob("MP_CV_THREADS", -1)
    torch.cuda.synchronize()
    for k, v in keep.items():
        assert torch.equal(torch.nan_to_num(r2[k]), torch.nan_to_num(v)), k
