#!/usr/bin/env python
"""Benchmark: batched SL path DP + ST speed DP plans/sec on MI355X.

One step = one pass of the hot path over one batch of synthetic scenarios that
already sit in HBM: the SL path DP over a 64-layer x 128-lane lattice with 50
obstacles per scenario (every layer relaxed, no early abort) followed by the
ST speed DP on a default ST lattice (11 time layers x 97 s cells, 5 moving
blocked intervals per layer), and -- with more than one GPU -- one RCCL
all-gather of the per-scenario results.  Scenarios are independent, so ranks
take disjoint contiguous blocks of scenarios and nothing else is exchanged.

    python bench.py [--gpus N] [--steps K] [--warmup W]

  N = 1: 4096 scenarios on one GPU.
  N > 1: 65536 scenarios split over the N GPUs (strong scaling); --scenarios P
         instead gives every GPU P scenarios (weak scaling).
With N > 1 and no launcher in the environment the script starts its own N ranks
(torch.distributed.run) before it touches the GPU.

The steps are timed twice (median of --reps timed blocks of K steps each).  First on
one stream with HIP events around every launch: the kernel times and rooflines.
Then with the batch cut into --streams parts (default 2), part k on stream k: SL
plan -> ST DP per part, no part waiting for another, so that the tail of one part's
kernel runs beside the other part's.  `value` is the schedule --streams names (two
parts by default, --streams 1 for one stream) whichever of them measured faster;
`value_one_stream` and `value_two_streams` are always both printed, and
`value_end_to_end` is the headline schedule with the host-to-device copy of the
inputs in front of every step and the results handed over behind it.

    python bench.py --gpus 1 --total-scenarios 65536   # the N = 1 anchor of the strong-scaling
                                                        # curve (the workload --gpus N > 1 splits)

Prints ONE JSON line (rank 0): plans/sec, the per-kernel rooflines, the same
workload with the host->device copies of its inputs inside the step, the CPU oracle
timed on a sample of the same batch (and its results compared with the GPU's), and
a short run of the whole-replan workload (--workload pipeline makes that the headline).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_S, N_L, N_OB = 64, 128, 50
ST_N_T, ST_N_S, ST_R = 11, 97, 5
SPEED_LIM = 25.0 * 5.0 / 18.0
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec
VALU_FP64_PEAK_TF = 78.6     # MI355X vector fp64 spec (counts a fused multiply-add as two)
VALU_FP64_NO_FMA_TF = 39.3   # the same pipes when no multiply-add may be fused (bit-exact Python arithmetic)
ISSUE_PEAK_GINST = 1024 * 2.4e9 / 4 / 1e9   # wave64 vector instructions/s: 1024 SIMDs, 4 cycles each, 2.4 GHz
ST_EDGES_PRICED_PER_PLAN = 977   # of 42 874, on the benchmark's batch (tools/count_st.py, counting build of the speed DP; profiles/r03g_st_counts.txt)
CFG4_TOTAL = 65536


def make_scenarios(first_id, B):
    """Synthetic batch: layers every 1.5 m up to 96 m, lanes linspace(-1.45, 1.45, 128),
    ego l ~ U(-0.3, 0.3), 50 obstacles s ~ U(2, 96), l ~ U(-2.5, 2.5), one radius ~ U(0.3, 0.8)
    per scenario; PCG64 seeded with 1000 + scenario id.  ST side: start speed ~ U(3, 8),
    five blocked intervals (s0 + v*t, s0 + v*t + 3) with randomly shifted s0."""
    s_layers = np.broadcast_to(1.5 * (np.arange(N_S) + 1.0), (B, N_S)).copy()
    l_row = np.broadcast_to(np.linspace(-1.45, 1.45, N_L), (B, N_L)).copy()
    ego = np.zeros((B, 2))
    ob_s, ob_l = np.zeros((B, N_OB)), np.zeros((B, N_OB))
    ob_dist = np.zeros(B)
    cur_vel = np.zeros(B)
    down = np.zeros((B, ST_N_T, ST_R))
    base = np.array([(12, 2), (20, 0), (6, 4), (25, -1), (15, 1)], dtype=np.float64)
    for b in range(B):
        rng = np.random.Generator(np.random.PCG64(1000 + first_id + b))
        ego[b, 1] = rng.uniform(-0.3, 0.3)
        ob_s[b] = rng.uniform(2, 96, N_OB)
        ob_l[b] = rng.uniform(-2.5, 2.5, N_OB)
        ob_dist[b] = rng.uniform(0.3, 0.8)
        cur_vel[b] = rng.uniform(3, 8)
        s0 = base[:, 0] * (96.0 / 28.0) + rng.uniform(-4, 4, ST_R)
        down[b] = s0[None, :] + base[None, :, 1] * 0.5 * np.arange(ST_N_T)[:, None]
    up = down + 3.0
    return dict(s_layers=s_layers, l_row=l_row, ego=ego, ob_s=ob_s, ob_l=ob_l,
                n_ob=np.full(B, N_OB, np.int32), ob_dist=ob_dist, cur_vel=cur_vel, down=down, up=up,
                n_reg=np.full((B, ST_N_T), ST_R, np.int32), st_n_s=np.full(B, ST_N_S, np.int32))


def make_scenes(B, n_st=192, n_dy=64, seed=5):
    """Full-pipeline workload: robot-frame scenes on curved reference lines y = c x^2 with
    16..21 knots every 4.5 m (map length <= 100 m), ego speed (U(3,8), U(-.5,.5)), 192
    static and 64 moving obstacles (radius 0.8, 11 predicted frames) spread along the
    line, most of them beside the road so that a share of the scenes keeps a free path."""
    rng = np.random.default_rng(seed)
    nk_max = 21
    kx, ky = np.zeros((B, nk_max)), np.zeros((B, nk_max))
    nk = rng.integers(16, nk_max + 1, B).astype(np.int32)
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
                dy_rad=np.full((B, n_dy), 0.8), n_dy=np.full(B, n_dy, np.int32), dy_vel=vel)


def to_world_frame(sc, seed=11):
    """The scenes of make_scenes as a caller of PlannerInterface.run_step holds them: a random
    ego pose (R, t) per scene, waypoints and obstacles (statics first, then the moving ones
    with their velocity) in the world frame.  Every obstacle lies ahead of the ego."""
    B = sc["knots_x"].shape[0]
    rng = np.random.default_rng(seed)
    th = rng.uniform(-np.pi, np.pi, B)
    rot = np.stack([np.stack([np.cos(th), -np.sin(th)], 1), np.stack([np.sin(th), np.cos(th)], 1)], 1)   # [B, 2, 2]
    trans = rng.uniform(-300, 300, (B, 2))

    def turn(p):        # R p for [B, n, 2]
        return np.einsum("bij,bnj->bni", rot, p)

    way = turn(np.stack([sc["knots_x"], sc["knots_y"]], 2)) + trans[:, None, :]
    ob_pos = turn(np.concatenate([sc["st_pos"], sc["dy_pos"][:, :, 0, :]], 1)) + trans[:, None, :]
    ob_vel = turn(np.concatenate([np.zeros_like(sc["st_pos"]), sc["dy_vel"]], 1))
    return dict(rot=rot.reshape(B, 4), trans=trans, ego_vel_w=np.einsum("bij,bj->bi", rot, sc["ego_vel"]),
                way_xy=way, n_way=sc["n_knots"].astype(np.int32), ob_pos=ob_pos, ob_vel=ob_vel,
                ob_rad=np.concatenate([sc["st_rad"], sc["dy_rad"]], 1),
                n_ob=(sc["n_st"] + sc["n_dy"]).astype(np.int32))


# ---------------------------------------------------------------------------
# CPU legs (the oracle is the checker and the reported baseline, never the product)
# ---------------------------------------------------------------------------
def cpu_pipeline(sc, n_sample, threads):
    """The CPU oracle's whole replan (Frenet map, SL DP, path sampling, ST map, ST DP,
    speed profile, conversion back) on the first n_sample scenes, `threads` at a time.
    Returns (plans/s, seconds, per-scene outcome: None = no plan, else the final path)."""
    import contextlib
    import io
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    orc.build()
    orc.lib()

    def one(b):
        nk = int(sc["n_knots"][b])
        st = [(p, r) for p, r in zip(sc["st_pos"][b], sc["st_rad"][b])]
        dy = [(fr, r) for fr, r in zip(sc["dy_pos"][b], sc["dy_rad"][b])]
        try:
            sl = orc.SLMap(sc["knots_x"][b, :nk], sc["knots_y"][b, :nk], 0.0, sc["ego_vel"][b], st, dy)
        except ValueError:
            return None
        found, curve, _ = orc.sl_plan(sl)
        if not found:
            return None
        pb = sl.path_sampling(curve)
        n_s, n_t = orc.st_dims(pb[-1][0])
        down, up, n_reg = orc.st_add_obstacle(n_t, pb, sl.st_ob_traj, sl.dy_ob_traj)
        sr = orc.st_find_speed(n_t, n_s, down, up, n_reg, sl.robot_vel, SPEED_LIM)
        ss, vv = orc.speed_profile(sr["path_s_ind"], sl.robot_vel)
        xy = sl.path_convert(sl.path_sampling(curve, ss))
        return np.asarray(xy, dtype=np.float64), np.asarray(vv, dtype=np.float64)

    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()), ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(one, range(n_sample)))
    dt = time.perf_counter() - t0
    return n_sample / dt, dt, res


def cpu_baseline(sc, n_sample, threads):
    """The CPU oracle (a quirk-exact C restatement of the reference planner) on the
    first n_sample scenarios of the same batch, `threads` scenarios at a time.
    Returns (plans/s, seconds, [(SL result, ST result), ...])."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    orc.build()
    orc.lib()

    def one(b):
        s_map = np.repeat(sc["s_layers"][b][:, None], N_L, axis=1)
        l_map = np.broadcast_to(sc["l_row"][b], (N_S, N_L))
        ob = np.stack([sc["ob_s"][b], sc["ob_l"][b]], 1)
        r = orc.sl_find_path(s_map, l_map, int((N_L - 1) / 2), sc["ego"][b], ob, sc["ob_dist"][b],
                             no_early_exit=True)
        s = orc.st_find_speed(ST_N_T, ST_N_S, sc["down"][b], sc["up"][b], sc["n_reg"][b], float(sc["cur_vel"][b]),
                              SPEED_LIM)
        best = float(r["cost_map"][-1, r["path_idx"][-1]]) if r["found"] else float("nan")
        return (dict(found=bool(r["found"]), path_idx=r["path_idx"].copy(), layers_done=int(r["layers_done"]),
                     best_cost=best), s["path_s_ind"].copy())

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(one, range(n_sample)))
    dt = time.perf_counter() - t0
    return n_sample / dt, dt, res


def single_plan_latencies(calls=200):
    """One plan per call, the reference's own use (one replan per tick): launch to synchronised
    result in microseconds (median of `calls` calls; the same call replayed as a HIP graph),
    for the default SL lattice with 3 static obstacles (3 x 5 nodes), the default ST grid with
    5 moving blocked intervals at 29 / 61 / 101 cells, and one whole replan of a scene with
    256 obstacles.  Beside each: the C oracle on one host thread, and the reference's own
    Python time as measured once in the build container (a constant; its source does not travel)."""
    import torch
    from motion_planning_amd import batch as mb, pipeline as pl
    from oracle import oracle as orc
    orc.build()
    orc.lib()

    def wall_us(fn, n):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return 1e6 * float(np.median(ts))

    def graph_us(fn, n):
        g = torch.cuda.CUDAGraph()
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            fn()
        return wall_us(g.replay, n)

    def cpu_us(fn, n):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return 1e6 * float(np.median(ts))

    res = {"calls": calls, "timing": "time.perf_counter around call + device synchronisation, median; graph: the "
                                     "same call captured once and replayed"}
    # SL: 7 waypoints every 4.5 m, 3 static obstacles of radius 1 (the reference's default lattice)
    s_layers = np.array([[8.0, 16.0, 28.0]])
    l_row = np.linspace(-1.45, 1.45, 5)[None]
    ob = np.array([[9.5, 0.2], [16.0, -1.0], [21.5, 1.2]])
    sb = mb.SLBatch(s_layers, l_row, np.zeros((1, 2)), ob[None, :, 0], ob[None, :, 1], np.array([3], np.int32),
                    np.array([1.0]))
    out = mb.SLPlanOut(1, 3, 5, want_maps=False)
    prm = mb.sl_params(n_l=5)
    mb.sl_plan(sb, prm, out)
    assert out.path_idx.cpu().tolist() == [[4, 4, 3]] and int(out.status[0]) == 1   # the reference's answer
    s_map, l_map = np.repeat(s_layers[0][:, None], 5, axis=1), np.broadcast_to(l_row[0], (3, 5))
    res["sl_default_3x5_3_obstacles"] = {
        "entry_point": "mp_sl_plan", "gpu_us": wall_us(lambda: mb.sl_plan(sb, prm, out), calls),
        "gpu_graph_us": graph_us(lambda: mb.sl_plan(sb, prm, out), calls),
        "c_oracle_1_thread_us": cpu_us(lambda: orc.sl_find_path(s_map, l_map, 2, np.zeros(2), ob, 1.0), 50),
        "reference_python_us": 62000.0 * 55 / 280,
        "reference_python_note": "222 us per edge (0.062 s for the 280 edges of a 12-layer default lattice) x the 55 "
                                 "edges of this 3-layer lattice"}
    # ST: 11 time layers, 5 moving intervals, three grid widths
    base = np.array([(12, 2), (20, 0), (6, 4), (25, -1), (15, 1)], dtype=np.float64)
    st = {}
    for s_end, ref_ms in ((28.0, 34.0), (60.0, 144.0), (100.0, 396.0)):
        n_s = int(np.floor(s_end / 1.0)) + 1
        s0 = base[:, 0] * (s_end / 28.0)
        down = (s0[None, :] + base[None, :, 1] * 0.5 * np.arange(ST_N_T)[:, None])[None]
        n_reg = np.full((1, ST_N_T), 5, np.int32)
        args = (mb.to_dev(np.array([n_s]), torch.int32), mb.to_dev(down), mb.to_dev(down + 3.0),
                mb.to_dev(n_reg, torch.int32), mb.to_dev(np.array([5.0])), SPEED_LIM)
        o = mb.st_dp(*args, want_tables=False, n_s_max=n_s)
        want = orc.st_find_speed(ST_N_T, n_s, down[0], down[0] + 3.0, n_reg[0], 5.0, SPEED_LIM)["path_s_ind"]
        got = o["path_s_ind"][0, :int(o["path_len"][0])].cpu().numpy()
        assert np.array_equal(got, want), "single ST plan differs from the oracle at %d cells" % n_s
        fn = lambda: mb.st_dp(*args, want_tables=False, n_s_max=n_s, out=o)
        st["%d_cells" % n_s] = {
            "gpu_us": wall_us(fn, calls), "gpu_graph_us": graph_us(fn, calls),
            "c_oracle_1_thread_us": cpu_us(lambda: orc.st_find_speed(ST_N_T, n_s, down[0], down[0] + 3.0, n_reg[0],
                                                                      5.0, SPEED_LIM), 5),
            "reference_python_us": 1e3 * ref_ms}
    res["st_default_5_intervals"] = dict(st, entry_point="mp_st_dp")
    # whole replan of one scene
    sc = make_scenes(1)
    dev = {k: mb.to_dev(v, torch.int32 if v.dtype == np.int32 else torch.float64) for k, v in sc.items()}
    dev["line_vec"] = torch.zeros(1, dtype=torch.float64, device="cuda")
    params = pl.default_params()
    dims = pl.plan_dims(1, sc["knots_x"].shape[1], sc["st_pos"].shape[1], sc["dy_pos"].shape[1], n_l=5, m_max=200,
                        params=params)
    arena = pl.PlanArena(dims)
    fn = lambda: pl.plan_scenes(dev, dims, params, arena)
    res["whole_replan_1_scene_256_obstacles"] = {
        "entry_point": "mp_plan_scenes", "gpu_us": wall_us(fn, calls), "gpu_graph_us": graph_us(fn, calls),
        "c_oracle_1_thread_us": 1e6 * cpu_pipeline(sc, 1, 1)[1],
        "reference_python_us": 0.6e6,
        "reference_python_note": "0.4-0.8 s for SLMap alone on such a scene"}
    return res


def check_lattice_parity(cpu_res, out, st, sc):
    """GPU outputs of the timed run against the oracle's, scenario by scenario: outcome,
    lane per layer, layers relaxed and end cost of the SL plan; the ST plan's s indices.
    A lane sequence may differ from the oracle's only where the two are a tie: the GPU's
    sequence is then priced edge by edge with the oracle's own edge cost, accumulated in
    the reference's order, and must come within 1e-9 relative of the oracle's end cost."""
    from oracle import oracle as orc
    n = len(cpu_res)
    status = out.status[:n].cpu().numpy()
    path = out.path_idx[:n].cpu().numpy()
    layers = out.layers_done[:n].cpu().numpy()
    best = out.best_cost[:n].cpu().numpy()
    st_path = st["path_s_ind"][:n].cpu().numpy()
    st_len = st["path_len"][:n].cpu().numpy()
    ties = 0
    for b, (r, sp) in enumerate(cpu_res):
        assert (status[b] == 1) == r["found"], "scenario %d: found differs from the oracle" % b
        assert int(layers[b]) == r["layers_done"], "scenario %d: layers relaxed differ" % b
        if r["found"]:
            assert abs(best[b] - r["best_cost"]) <= 1e-9 * abs(r["best_cost"]), "scenario %d: end cost differs" % b
            if not np.array_equal(path[b], r["path_idx"]):
                s_map = np.repeat(sc["s_layers"][b][:, None], N_L, axis=1)
                l_map = np.broadcast_to(sc["l_row"][b], (N_S, N_L))
                ob = np.stack([sc["ob_s"][b], sc["ob_l"][b]], 1)
                cost = 0.0
                for s in range(N_S):
                    cost = orc.sl_cal_cost(s_map, l_map, int((N_L - 1) / 2), sc["ego"][b], ob, sc["ob_dist"][b],
                                           s - 1, int(path[b, s - 1]) if s else 0, s, int(path[b, s]))[0] + cost
                assert abs(cost - r["best_cost"]) <= 1e-9 * abs(r["best_cost"]), \
                    "scenario %d: lane sequence differs from the oracle's and is no tie (%r against %r)" \
                    % (b, cost, r["best_cost"])
                ties += 1
        else:
            assert (path[b] == -1).all() and np.isnan(best[b])
        assert int(st_len[b]) == len(sp) and np.array_equal(st_path[b, :len(sp)], sp), \
            "scenario %d: ST path differs from the oracle" % b
    return ties


# ---------------------------------------------------------------------------
# launch plumbing
# ---------------------------------------------------------------------------
def exchange_on(world):
    """The result exchange runs with more than one rank; MP_BENCH_FORCE_DIST=1 also runs it
    (process group, barrier, all-gather, max over ranks) in a one-rank job, which is how the
    RCCL calls are rehearsed on a one-GPU box."""
    return world > 1 or os.environ.get("MP_BENCH_FORCE_DIST") == "1"


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks ourselves, before this process has
    touched a GPU (counting devices does not initialise one), and hand on their exit code."""
    import torch
    env = dict(os.environ)
    n_dev = torch.cuda.device_count()
    if args.launch_check:
        env["MP_BENCH_BACKEND"] = "gloo"
    elif n_dev < args.gpus and "MP_BENCH_BACKEND" not in env:
        sys.stderr.write("bench.py: %d GPU(s) visible for --gpus %d: ranks share the card(s) and the gather runs "
                         "over gloo (a rehearsal, not a scaling measurement)\n" % (n_dev, args.gpus))
        env["MP_BENCH_BACKEND"] = "gloo"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


_STREAMS = {}


def shared_stream(name):
    """One stream per role for the whole run ("part0", "part1", ..., "copy"): a process has
    four hardware queues, and streams beyond them share a queue with another stream and
    wait for its work."""
    import torch
    if name not in _STREAMS:
        _STREAMS[name] = torch.cuda.Stream()
    return _STREAMS[name]


class ResultGather:
    """The per-step exchange of result records when more than one GPU takes part: pack on
    the device (one kernel), then one all-gather issued asynchronously on the process
    group's own stream into one of two buffer pairs, so that the exchange of batch k runs
    while the kernels of batch k+1 do; a buffer pair is waited for before it is reused,
    and drain() waits for everything (called before the clock stops).  Shards are padded
    to the largest one (rows), so ranks may hold different numbers of scenarios."""

    def __init__(self, world, backend, B, rows, n_s, n_t, device="cuda", active=None):
        import torch
        from motion_planning_amd import dist as mdist
        self.world, self.backend, self.mdist, self.B, self.rows = world, backend, mdist, B, rows
        self.k = 0
        self.active = exchange_on(world) if active is None else bool(active)   # False: the no-exchange leg
        if not self.active:
            return
        width = mdist.record_width(n_s, n_t)
        gdev = "cuda" if backend == "nccl" else "cpu"
        self.packed = [torch.zeros((rows, width), dtype=torch.uint8, device=device) for _ in range(2)]
        self.gathered = [torch.empty((world * rows, width), dtype=torch.uint8, device=gdev) for _ in range(2)]
        self.pending = [None, None]

    def submit(self, path_idx, status, layers_done, best_cost, st_path, st_len):
        if not self.active:
            return
        i = self.k & 1
        if self.pending[i] is not None:
            self.pending[i].wait()
        self.mdist.pack_results(path_idx, status, layers_done, best_cost, st_path, st_len, out=self.packed[i][:self.B])
        src = self.packed[i] if self.backend == "nccl" else self.packed[i].cpu()
        self.pending[i] = self.mdist.gather_results_same_size(src, self.gathered[i], async_op=True)
        self.k += 1

    def submit_parts(self, parts):
        """The same exchange for a batch whose parts were planned on streams of their own:
        parts = [(lo, hi, stream, (path_idx, status, layers_done, best_cost, st_path, st_len))].
        Each part packs its rows on its own stream; the collective waits for every part,
        no part waits for another."""
        if not self.active:
            return
        import torch
        main = torch.cuda.current_stream()
        i = self.k & 1
        if self.pending[i] is not None:
            self.pending[i].wait()             # the collective that read packed[i] two steps ago
            self.pending[i] = None
            freed = torch.cuda.Event()
            freed.record(main)
            for _lo, _hi, stream, _o in parts:
                stream.wait_event(freed)
        for lo, hi, stream, o in parts:
            with torch.cuda.stream(stream):
                self.mdist.pack_results(*o, out=self.packed[i][lo:hi])
                ev = torch.cuda.Event()
                ev.record(stream)
            main.wait_event(ev)
        src = self.packed[i] if self.backend == "nccl" else self.packed[i].cpu()
        self.pending[i] = self.mdist.gather_results_same_size(src, self.gathered[i], async_op=True)
        self.k += 1

    def drain(self):
        if not self.active:
            return
        for i in range(2):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None

    def last(self):
        return self.gathered[(self.k - 1) & 1]

    def rows_of(self, rank, count):
        """Rank `rank`'s records of the last exchange (its first `count` rows)."""
        return self.last()[rank * self.rows:rank * self.rows + count]


class Job:
    """Rank, world, backend and the shard of this process."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        # MP_BENCH_BACKEND=gloo rehearses the multi-rank path on a one-GPU box (ranks
        # share the card and the gather is staged through host memory)
        self.backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
        self.gpu = not args.launch_check
        if self.gpu:
            torch.cuda.set_device(local % torch.cuda.device_count())
        self.dist_on = exchange_on(self.world)
        if self.dist_on:
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)   # a forced one-rank job started without a launcher
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
            else:
                dist.init_process_group(self.backend)
        # workload size: per-GPU count (weak) or a total split over the ranks (strong)
        from motion_planning_amd import dist as mdist
        if args.total_scenarios is None and args.scenarios is None:
            if self.world > 1:
                args.total_scenarios = CFG4_TOTAL
            else:
                args.scenarios = 4096
        if args.total_scenarios is not None:
            self.scaling = "strong"
            self.total = args.total_scenarios
            self.lo, self.hi = mdist.shard_range(self.total, self.rank, self.world)
            self.rows = mdist.shard_range(self.total, 0, self.world)[1]   # rank 0 holds a largest shard
        else:
            self.scaling = "weak"
            self.total = args.scenarios * self.world
            self.lo, self.hi = self.rank * args.scenarios, (self.rank + 1) * args.scenarios
            self.rows = args.scenarios
        self.B = self.hi - self.lo
        if self.B <= 0:
            raise SystemExit("rank %d has no scenarios (total %d over %d ranks)" % (self.rank, self.total, self.world))

    def describe(self):
        """What the record needs to show that N ranks ran on N GPUs: group size and backend as
        torch.distributed reports them, the collective library's version, and every rank's
        device (index, name, PCI bus id) gathered to all ranks."""
        import torch
        import torch.distributed as dist
        me = {"rank": self.rank, "pid": os.getpid(), "host": socket.gethostname()}
        if self.gpu:
            i = torch.cuda.current_device()
            pr = torch.cuda.get_device_properties(i)
            me.update(device_index=i, device_name=pr.name,
                      pci_bus_id="%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xff,
                                                     getattr(pr, "pci_device_id", 0)),
                      uuid=str(getattr(pr, "uuid", "")))
        d = {"world_size": self.world, "backend": None, "collective_library": None, "ranks": [me]}
        if self.dist_on:
            d["world_size"] = dist.get_world_size()
            d["backend"] = dist.get_backend()
            if self.backend == "nccl":
                try:
                    d["collective_library"] = "RCCL (torch.cuda.nccl.version) %s" % ".".join(
                        str(v) for v in torch.cuda.nccl.version())
                except Exception as e:   # version query is informational only
                    d["collective_library"] = "unavailable: %s" % e
            ranks = [None] * d["world_size"]
            dist.all_gather_object(ranks, me)
            d["ranks"] = ranks
            devs = {(r.get("host"), r.get("pci_bus_id", r.get("pid"))) for r in ranks}
            d["distinct_devices"] = len(devs) if self.gpu else 0
        return d

    def fence(self):
        import torch
        import torch.distributed as dist
        if self.gpu:
            torch.cuda.synchronize()
        if self.dist_on:
            dist.barrier()
        if self.gpu:
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        import torch
        import torch.distributed as dist
        if not self.dist_on:
            return seconds
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor(seconds, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def close(self):
        import torch.distributed as dist
        if self.dist_on:
            dist.destroy_process_group()


def timed_blocks(job, step, drain, steps, reps):
    """`reps` timed blocks of exactly `steps` steps, each bracketed by barrier + device
    synchronisation on both sides; returns the blocks' seconds (max over ranks)."""
    import gc
    secs = []
    gc_was_on = gc.isenabled()
    gc.disable()   # a collection inside a block of a few milliseconds shows up as a block several times as long
    try:
        for _ in range(reps):
            job.fence()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            drain()
            job.fence()
            secs.append(time.perf_counter() - t0)
    finally:
        if gc_was_on:
            gc.enable()
    return job.max_over_ranks(secs)


def load_pmc(name="pmc_latest.json"):
    """Per-kernel counter means (per launch) from the last committed rocprofv3 --pmc passes of this
    same command; bench.py cannot collect them itself."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            d = json.load(f)
        k = dict(d.get("kernels", {}))
        k["_source"] = d.get("source")
        k["_scenarios"] = d.get("scenarios", 4096)
        return k
    except (OSError, ValueError):
        return {}


def launch_check(job):
    """--launch-check: the launcher, the sharding and the result exchange without any
    kernel (CPU tensors, gloo), so that the multi-rank plumbing of this script can be
    exercised where no GPU is visible.  Every rank builds the records its scenarios would
    produce from their ids, exchanges them twice through ResultGather and checks that it
    holds every scenario's record in scenario order."""
    import torch
    from motion_planning_amd import dist as mdist
    n_s, n_t = 6, ST_N_T

    def fake(lo, hi):
        ids = torch.arange(lo, hi, dtype=torch.int32)
        path = (ids[:, None] * 7 + torch.arange(n_s, dtype=torch.int32)[None]) % 128
        status = (ids % 3 != 0).to(torch.int32)
        layers = torch.full_like(ids, n_s)
        best = ids.double() * 1.25 + 0.1
        st_len = (ids % n_t + 1).to(torch.int32)
        st_path = ((ids[:, None] + torch.arange(n_t, dtype=torch.int32)[None]) % 97).to(torch.int32)
        return path, status, layers, best, st_path, st_len

    gather = ResultGather(job.world, job.backend, job.B, job.rows, n_s, n_t, device="cpu")
    mine = fake(job.lo, job.hi)
    shards = []
    for _ in range(2):
        gather.submit(*mine)
    gather.drain()
    job.fence()
    for r in range(job.world):
        lo, hi = (mdist.shard_range(job.total, r, job.world) if job.scaling == "strong"
                  else (r * job.B, (r + 1) * job.B))
        shards.append(hi - lo)
        if gather.active:
            u = mdist.unpack_results(gather.rows_of(r, hi - lo), n_s, n_t)
            want = fake(lo, hi)
            assert torch.equal(u["path_idx"], want[0]) and torch.equal(u["status"], want[1])
            assert torch.equal(u["best_cost"], want[3]) and torch.equal(u["st_path"], want[4])
    # the fields a multi-rank bench line carries about the exchange and the ranks, from the same
    # code paths (no kernels here: a "step" is the exchange of the fake records alone)
    gather_off = ResultGather(job.world, job.backend, job.B, job.rows, n_s, n_t, device="cpu", active=False)
    sec_on = float(np.median(timed_blocks(job, lambda i: gather.submit(*mine), gather.drain, 3, 2)))
    sec_off = float(np.median(timed_blocks(job, lambda i: gather_off.submit(*mine), gather_off.drain, 3, 2)))
    who = job.describe()
    if job.rank == 0:
        v, v_off = job.total * 3 / sec_on, job.total * 3 / sec_off
        print(json.dumps({"launch_check": "ok", "n_gpus": job.world, "total_scenarios": job.total,
                          "scaling": job.scaling, "shards": shards, "backend": job.backend if job.dist_on else None,
                          "config": {"per_gpu_batch": job.B, "total_scenarios": job.total},
                          "value": v, "value_no_exchange": v_off, "exchange_overhead_frac": 1.0 - v / v_off,
                          "scenarios_rank0": job.B, "job": who}))


# ---------------------------------------------------------------------------
# whole-replan workload
# ---------------------------------------------------------------------------
def bench_pipeline(args, job, steps, warmup, reps, cpu_sample):
    """One step = one mp_plan_scenes call over B scenes resident in HBM (every stage of the
    replan) and, with more than one GPU, the result gather.  Returns the result dict."""
    import torch
    from motion_planning_amd import batch as mb, dist as mdist, pipeline as pl

    B = args.pipeline_scenes
    world, rank = job.world, job.rank
    sc = make_scenes(B, seed=5 + rank)
    # the scenes as PlannerInterface.run_step receives them: world frame, an ego pose per scene
    wsc = to_world_frame(sc, seed=11 + rank)
    wdev = {k: mb.to_dev(v, torch.int32 if v.dtype == np.int32 else torch.float64) for k, v in wsc.items()}
    n_st_max, n_dy_max = sc["st_pos"].shape[1], sc["dy_pos"].shape[1]
    ignore = 1e9   # RobotMap's 20 m obstacle range lifted: every obstacle along the line takes part
    params = pl.default_params()
    dims = pl.plan_dims(B, sc["knots_x"].shape[1], n_st_max, n_dy_max, n_l=5, m_max=200, params=params, world=True)
    arena = pl.PlanArena(dims)
    a = arena.arrays
    gather = ResultGather(world, job.backend, B, B, dims.n_s_max, dims.n_t)

    def step(_i=None):
        pl.plan_world_scenes(wdev, dims, params, arena, ignore_dist=ignore)
        gather.submit(a["path_idx"], a["sl_status"], a["layers_done"], a["best_cost"], a["st_path_s_ind"],
                      a["st_path_len"])

    for _ in range(warmup):
        step()
    gather.drain()
    secs = timed_blocks(job, step, gather.drain, steps, reps)
    if job.dist_on:
        u = mdist.unpack_results(gather.rows_of(rank, B).to("cuda"), dims.n_s_max, dims.n_t)
        assert torch.equal(u["path_idx"], a["path_idx"]) and torch.equal(u["st_path"], a["st_path_s_ind"])
    # the same steps with the scenes cut into parts, each part's whole chain on a stream of
    # its own with an arena of its own (see SplitLoop): eight launches per part, and the
    # tail of every one of them runs beside the other part's kernels
    split_secs = None
    n_parts = args.streams
    if n_parts > 1 and B >= 64 * n_parts:
        parts = []
        for k in range(n_parts):
            lo, hi = mdist.shard_range(B, k, n_parts)
            pdims = pl.plan_dims(hi - lo, sc["knots_x"].shape[1], n_st_max, n_dy_max, n_l=5, m_max=200, params=params,
                                 world=True)
            parts.append(dict(lo=lo, hi=hi, dev={kk: v[lo:hi].contiguous() for kk, v in wdev.items()}, dims=pdims,
                              arena=pl.PlanArena(pdims), stream=shared_stream("part%d" % k)))
        torch.cuda.synchronize()

        def split_step(_i=None):
            for p in parts:
                with torch.cuda.stream(p["stream"]):
                    pl.plan_world_scenes(p["dev"], p["dims"], params, p["arena"], ignore_dist=ignore)
            gather.submit_parts([(p["lo"], p["hi"], p["stream"],
                                  tuple(p["arena"].arrays[kk] for kk in ("path_idx", "sl_status", "layers_done",
                                                                         "best_cost", "st_path_s_ind", "st_path_len")))
                                 for p in parts])

        for _ in range(max(2, warmup)):
            split_step()
        gather.drain()
        split_secs = timed_blocks(job, split_step, gather.drain, steps, reps)
        torch.cuda.synchronize()
        for kk in ("status", "n_final", "path_idx", "st_path_s_ind", "final_x", "final_y", "vv", "world_x", "world_y"):
            got = torch.cat([p["arena"].arrays[kk] for p in parts])
            assert torch.equal(got, a[kk]) or (got.is_floating_point() and
                                               torch.equal(got.nan_to_num(), a[kk].nan_to_num())), kk
        if job.dist_on:
            u = mdist.unpack_results(gather.rows_of(rank, B).to("cuda"), dims.n_s_max, dims.n_t)
            assert torch.equal(u["path_idx"], a["path_idx"]) and torch.equal(u["st_path"], a["st_path_s_ind"])
    status = a["status"].cpu().numpy()
    assert set(np.unique(status)) <= {1, 0, -4}, "unexpected scene status %s" % np.unique(status)
    n_final = a["n_final"].cpu().numpy()
    assert (n_final[status == 1] >= 2).all()
    # every obstacle of every scene was kept by the ingestion, in its class
    assert torch.equal(a["w_n_st"].cpu(), torch.as_tensor(sc["n_st"])) and torch.equal(a["w_n_dy"].cpu(),
                                                                                       torch.as_tensor(sc["n_dy"]))
    # the chain alone, from the robot-frame scene the ingestion left in the arena (what this leg
    # timed before it started from world-frame scenes)
    rdims = pl.plan_dims(B, sc["knots_x"].shape[1], n_st_max, n_dy_max, n_l=5, m_max=200, params=params)
    rarena = pl.PlanArena(rdims)
    rdev = dict(knots_x=a["w_knots_x"], knots_y=a["w_knots_y"], n_knots=wdev["n_way"], ego_vel=a["w_ego_vel"],
                st_pos=a["w_st_pos"], st_rad=a["w_st_rad"], n_st=a["w_n_st"], dy_pos=a["w_dy_pos"],
                dy_rad=a["w_dy_rad"], n_dy=a["w_n_dy"], line_vec=torch.zeros(B, dtype=torch.float64, device="cuda"))
    pl.plan_scenes(rdev, rdims, params, rarena)
    robot_secs = timed_blocks(job, lambda i: pl.plan_scenes(rdev, rdims, params, rarena), lambda: None, steps,
                              max(1, reps // 2))
    for kk in ("status", "n_final", "final_x", "final_y", "vv"):
        assert torch.equal(rarena.arrays[kk].nan_to_num(), a[kk].nan_to_num()), kk
    # per-stage durations from HIP events between the stages (three calls, averaged)
    stage = None
    for _ in range(3):
        _, ms = pl.plan_world_scenes_timed(wdev, dims, params, arena, ignore_dist=ignore)
        stage = ms if stage is None else [(n, t0 + t1) for (n, t0), (_, t1) in zip(stage, ms)]
    stage = [(n, t / 3.0) for n, t in stage]
    if rank != 0:
        return None
    one_sec = float(np.median(secs))
    one_secs = secs
    if split_secs:   # the schedule --streams names is the headline, whichever measured faster
        secs = split_secs
    sec = float(np.median(secs))
    ms = 1e3 * sec / steps
    stage_ms = dict(stage)
    # algorithmic work of the three dominant stages, as the reference writes it
    n_t = dims.n_t
    st_n_s = a["st_n_s"].cpu().numpy().astype(np.float64)
    n_reg = a["st_n_reg"].cpu().numpy().astype(np.float64)          # [B, n_t]
    ok = status == 1
    per_layer_edges = st_n_s * (st_n_s + 1) / 2.0                    # layers 2..n_t-1
    flop_st = float(np.sum(ok * (st_n_s * (30 + 9 * n_reg[:, 1]) +
                                 np.sum(per_layer_edges[:, None] * (30 + 9 * n_reg[:, 2:]), axis=1))))
    M = a["tab_m"].cpu().numpy().astype(np.float64)
    pts = sc["n_st"] + float(dims.n_frames) * sc["n_dy"] + 1
    flop_fe = float(np.sum((status != -5) * pts * (7 * M + 40)))
    pmc = load_pmc("pmc_pipeline_latest.json")

    def stage_roof(name, kernel, flop, note):
        t = stage_ms[name] * 1e-3
        r = {"kernel": kernel, "stage": name, "kernel_ms": stage_ms[name], "bound": "valu",
             "achieved": flop / t / 1e12, "peak": VALU_FP64_PEAK_TF, "unit": "TFLOP/s",
             "frac": flop / t / 1e12 / VALU_FP64_PEAK_TF, "algorithmic_flop_per_launch": flop, "note": note}
        k = pmc.get(kernel, {})
        if "SQ_INSTS_VALU" in k and pmc.get("_scenarios") == B:
            r["issue_slots_used"] = k["SQ_INSTS_VALU"] / t / 1e9 / ISSUE_PEAK_GINST
        return r

    # (the speed DP plans the nearest cells first and bounds the rest of the grid by their result, so
    # most edges are never priced and the reference's flops over its time are not a utilisation: its
    # entry is priced by its own vector instructions, like the interval kernel's below)
    st_roof = stage_roof("ST speed DP", "st_dp_kernel", flop_st,
                         "bound by vector instruction issue: frac = the kernel's own vector instructions "
                         "(SQ_INSTS_VALU of the committed counter pass) / duration against 1024 SIMDs x 2.4 GHz / 4 "
                         "cycles.  reference_work_tflops = (30 + 9 R) flop per ST edge over the scenes that reach "
                         "the speed search, R = that layer's interval count, / duration: what the reference would "
                         "have had to do, not what the kernel does (two sweeps, exact pruning).  The launch is "
                         "st_dp_plain_kernel, the build of st_dp_kernel's body for large batches (every relaxation "
                         "path, no tables, five waves per SIMD)")
    st_roof.update(bound="valu_issue", unit="G wave-instructions/s", peak=ISSUE_PEAK_GINST,
                   reference_work_tflops=st_roof["achieved"], frac=st_roof.get("issue_slots_used"),
                   achieved=st_roof["issue_slots_used"] * ISSUE_PEAK_GINST if st_roof.get("issue_slots_used") else None)
    roofs = [
        st_roof,
        stage_roof("front end (Frenet map, obstacle list)", "plan_frontend_kernel", flop_fe,
                   "(7 M + 40) flop per point to Frenet, M = the scene's reference-line samples; the kernel "
                   "compares only the samples of the blocks whose bounding box can hold the nearest one"),
    ]
    # reference-line tables: the reference integrates the arc length of every table entry from
    # the origin anew (Curve.calc_arc_len): one integrand value sqrt(1 + y'^2) per d_x of arc
    # plus one per spline piece crossed and the end point, ~19 flop each (quintic derivative in
    # power form, square, add, root, running sum); twice, for the Frenet and the inverse table
    n_kn = wsc["n_way"].astype(np.float64)
    k_ori = np.clip(np.rint((a["origin"][:, 0].cpu().numpy() - a["tab_x"][:, 0].cpu().numpy()) / params.d_x), 0, None)
    after = np.clip(M - k_ori, 0, None)                     # entries at or beyond the origin
    evals = after * (after + 1) / 2.0 + after * (1.0 + (n_kn - 1) / 2.0)
    flop_rl = float(np.sum((status != -5) * 2.0 * 19.0 * evals))
    # blocked intervals: every path point against every obstacle on every time layer, 7 flop a test
    n_path = a["n_path"].cpu().numpy().astype(np.float64)
    flop_reg = float(np.sum(7.0 * n_t * (sc["n_st"] + sc["n_dy"]) * n_path))
    roofs.append(stage_roof("reference-line tables", "refline_tables_kernel", flop_rl,
                            "19 flop per arc-length integrand value, (k - k_origin) + pieces crossed + 1 values for "
                            "table entry k as the reference integrates it from the origin anew, for the Frenet and "
                            "the inverse table; the kernel shares the integrand values among the entries"))
    # (the interval kernel scans a static obstacle once instead of on every time layer and drops
    # obstacles that cannot reach the path's bounding box, so the reference's test count over its
    # time would exceed the vector peak: its entry is priced by its own vector instructions)
    reg = stage_roof("ST blocked intervals", "st_regions_once_kernel", flop_reg,
                     "bound by vector instruction issue: frac = the kernel's own vector instructions "
                     "(SQ_INSTS_VALU of the committed counter pass) / duration against 1024 SIMDs x 2.4 GHz / 4 "
                     "cycles.  reference_work_tflops = 7 flop per distance test x n_t x (statics + moving) x path "
                     "points as the reference scans them / duration: not a utilisation (the kernel scans a static "
                     "obstacle once and drops obstacles that cannot reach the path's bounding box)")
    reg.update(bound="valu_issue", unit="G wave-instructions/s", peak=ISSUE_PEAK_GINST,
               reference_work_tflops=reg["achieved"], frac=reg.get("issue_slots_used"),
               achieved=reg["issue_slots_used"] * ISSUE_PEAK_GINST if reg.get("issue_slots_used") else None)
    roofs.append(reg)
    k = pmc.get("sl_plan_narrow_kernel", {})
    t_sl = stage_ms["SL path DP"] * 1e-3
    sl_entry = {"kernel": "sl_plan_narrow_kernel", "stage": "SL path DP", "kernel_ms": stage_ms["SL path DP"],
                "bound": "valu_issue", "unit": "G wave-instructions/s", "peak": ISSUE_PEAK_GINST,
                "note": "edge costs are not materialised and the per-sample sums are collapsed into per-layer "
                        "constants, so reference flops per edge do not describe this kernel; its bound is vector "
                        "instruction issue"}
    if "SQ_INSTS_VALU" in k and pmc.get("_scenarios") == B:
        sl_entry["achieved"] = k["SQ_INSTS_VALU"] / t_sl / 1e9
        sl_entry["frac"] = sl_entry["achieved"] / ISSUE_PEAK_GINST
    roofs.append(sl_entry)
    n_pts = B * (sc["st_pos"].shape[1] + sc["dy_pos"].shape[1] * 11 + 1)
    res = {
        "metric": "full replan plans/sec (world-frame scene in: ego-frame transform and obstacle filtering + "
                  "Cartesian->Frenet + SL path DP + ST speed DP + speed profile + path in the robot and world frames)",
        "value": world * B * steps / sec, "unit": "plans/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": ms, "timed_reps": reps,
        "ms_per_step_min": 1e3 * min(secs) / steps, "ms_per_step_max": 1e3 * max(secs) / steps,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "full pipeline: %d scenes/GPU, curved reference lines of 16-21 knots, 192 static + "
                               "64 moving obstacles (897 points to Frenet per scene), default lattice (5 lanes, "
                               "layers every 8 m), default ST grid; scenes given in the world frame with an ego "
                               "pose each, one mp_plan_world_scenes call per step" % B,
                   "scenarios_per_gpu": B, "n_l": 5, "n_st": 192, "n_dy": 64,
                   "parallelism": "scenario-sharded x%d" % world},
        "schedule": ("the scenes in %d parts on %d streams, one mp_plan_world_scenes call per part and step, parts "
                     "never wait for one another" % (n_parts, n_parts)) if split_secs
                    else "one stream: one mp_plan_world_scenes call over all scenes per step",
        "value_one_stream": world * B * steps / one_sec, "ms_per_step_one_stream": 1e3 * one_sec / steps,
        "value_two_streams": world * B * steps / float(np.median(split_secs)) if split_secs else None,
        "ms_per_step_one_stream_min": 1e3 * min(one_secs) / steps,
        "ms_per_step_one_stream_max": 1e3 * max(one_secs) / steps,
        "value_robot_frame_one_stream": world * B * steps / float(np.median(robot_secs)),
        "ms_per_step_robot_frame_one_stream": 1e3 * float(np.median(robot_secs)) / steps,
        "robot_frame": "mp_plan_scenes on the robot-frame scenes the ingestion left in the arena: the chain "
                       "without its first and last step, one stream, same results bit for bit",
        "stage_times": "stage_ms and the rooflines: HIP events between the stages of one-stream calls over all scenes",
        "c2f_points_per_sec": world * n_pts * steps / sec,
        "status_fractions": {"planned": float((status == 1).mean()), "no_path": float((status == 0).mean()),
                             "map_longer_than_100m": float((status == -4).mean())},
        "stage_ms": stage_ms, "stage_ms_sum": float(sum(stage_ms.values())),
        "roofline": max(roofs[:2], key=lambda r: r["kernel_ms"]),
        "rooflines": roofs, "pmc_source": pmc.get("_source"),
    }
    if cpu_sample > 0 and world == 1:  # the CPU leg belongs to the one-GPU run only
        threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        n = min(cpu_sample, B)
        # the oracle plans the robot-frame scenes the GPU's ingestion produced (that kernel is held
        # bit-equal to the reference's RobotMap by the world-frame fixtures of the tests); the way
        # back to the world frame is checked against R p + t
        sc_in = {k: a[w][:n].cpu().numpy() for k, w in (("knots_x", "w_knots_x"), ("knots_y", "w_knots_y"),
                                                         ("ego_vel", "w_ego_vel"), ("st_pos", "w_st_pos"),
                                                         ("st_rad", "w_st_rad"), ("dy_pos", "w_dy_pos"),
                                                         ("dy_rad", "w_dy_rad"))}
        sc_in["n_knots"] = wsc["n_way"][:n]
        v, dt, cres = cpu_pipeline(sc_in, n, threads)
        fx, fy, vv, gx, gy = (a[k][:n].cpu().numpy() for k in ("final_x", "final_y", "vv", "world_x", "world_y"))
        rot, trans = wsc["rot"].reshape(-1, 2, 2), wsc["trans"]
        for b, r in enumerate(cres):
            assert (r is not None) == (status[b] == 1), "scene %d: planned/not planned differs from the oracle" % b
            if r is not None:
                m = len(r[1])
                assert m == n_final[b], "scene %d: number of path points differs" % b
                np.testing.assert_allclose(np.stack([fx[b, :m], fy[b, :m]], 1), r[0], rtol=1e-9, atol=1e-8)
                np.testing.assert_allclose(vv[b, :m], r[1], rtol=1e-9, atol=1e-12)
                np.testing.assert_allclose(np.stack([gx[b, :m], gy[b, :m]], 1), r[0] @ rot[b].T + trans[b], rtol=1e-9,
                                           atol=1e-7)
        res["cpu_baseline"] = {"value": v, "unit": "plans/s", "cores": threads, "kind": "port",
                               "sample": "first %d scenes of the same batch, whole replan from the robot-frame scene "
                                         "on, C oracle driven from Python, %d threads, %.1f s" % (n, threads, dt),
                               "parity_checked": n}
    return res


# ---------------------------------------------------------------------------
# lattice workload
# ---------------------------------------------------------------------------
def materialised_roofline(sc, n, prm):
    """The two-kernel form of the same SL plan on the first n scenarios, outside the timed
    region: mp_sl_edge_cost writes every edge cost to HBM (fp64, 8 B/edge) and mp_sl_dp
    streams that tensor once -- the HBM-bound relaxation the fused kernel makes unnecessary,
    measured so that an HBM roofline fraction is on record."""
    import torch
    from motion_planning_amd import batch as mb
    sb = mb.SLBatch(sc["s_layers"][:n], sc["l_row"][:n], sc["ego"][:n], sc["ob_s"][:n], sc["ob_l"][:n],
                    sc["n_ob"][:n], sc["ob_dist"][:n])
    edge, _ = mb.sl_edge_cost(sb, prm)
    out = mb.sl_dp(edge, prm)
    torch.cuda.synchronize()
    assert torch.equal(out.path_idx[:, 0] >= 0, out.status == 1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    reps, t_e, t_d = 3, 0.0, 0.0
    for _ in range(reps):
        ev[0].record()
        edge, _ = mb.sl_edge_cost(sb, prm)
        ev[1].record()
        mb.sl_dp(edge, prm, out=out)
        ev[2].record()
        torch.cuda.synchronize()
        t_e += ev[0].elapsed_time(ev[1]) / reps
        t_d += ev[1].elapsed_time(ev[2]) / reps
    nbytes = 8.0 * n * N_S * N_L * N_L
    gbs = nbytes / (t_d * 1e-3) / 1e9
    # counters of the committed passes were taken on the default sample (1024 scenarios); the
    # kernel reads 16 bytes per lane, for which FETCH_SIZE on gfx950 reports half of the bytes
    k = load_pmc().get("sl_dp_kernel", {})
    traffic = None
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k and n == 1024:
        traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
    del edge
    return {"bound": "hbm", "kernel": "sl_dp_kernel (relaxation over the materialised fp64 edge tensor)",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "kernel_ms": t_d, "algorithmic_bytes_per_launch": nbytes, "scenarios": n,
            "edge_kernel": {"kernel": "sl_edge_cost_kernel", "kernel_ms": t_e,
                            "written_GBps": nbytes / (t_e * 1e-3) / 1e9,
                            "edges_evaluated_per_sec": n * N_S * N_L * N_L / (t_e * 1e-3)},
            "note": "not part of the timed step: the product path is the fused kernel; this is the same plan "
                    "through mp_sl_edge_cost + mp_sl_dp, whose DP kernel reads each edge cost once from HBM"}


class H2DLoop:
    """The same step with the host->device copy of its inputs inside: pinned host buffers,
    two sets of device buffers, copies on a stream of their own so that the copy of step
    i+1 runs beside the kernels of step i.  With n_parts > 1 the kernels run as in
    SplitLoop, every part over its rows of the one copied buffer."""

    SL_KEYS = (("s_layers", np.float64), ("l_row", np.float64), ("ego", np.float64), ("ob_s", np.float64),
               ("ob_l", np.float64), ("n_ob", np.int32), ("ob_dist", np.float64))
    ST_KEYS = (("st_n_s", np.int32), ("down", np.float64), ("up", np.float64), ("n_reg", np.int32),
               ("cur_vel", np.float64))

    def __init__(self, sc, prm, gather, n_parts=1, to_host=False):
        import torch
        from motion_planning_amd import batch as mb, dist as mdist
        self.torch, self.mb, self.mdist, self.prm, self.gather = torch, mb, mdist, prm, gather
        B = sc["s_layers"].shape[0]
        # to_host: the step also hands its results over -- through the exchange where one runs,
        # else as packed records that the packing kernel of each part writes into pinned host memory
        self.rec_dev = self.rec_host = None
        if to_host and not gather.active:
            width = mdist.record_width(N_S, ST_N_T)
            self.rec_dev = True
            self.rec_host = [torch.zeros((B, width), dtype=torch.uint8).pin_memory() for _ in range(2)]
        # n_parts > 1: still one copy per step, then the kernels of part k on stream k over
        # that part's rows of the copied arrays (parts wait for the copy, not for one another)
        self.ranges = [mdist.shard_range(B, k, n_parts) for k in range(n_parts)]
        # every input array in ONE pinned staging buffer (256-byte aligned slots), so that a
        # step is one copy, the way a caller with a staging arena hands a batch over
        arrs = {k: np.ascontiguousarray(sc[k], dtype=dt) for k, dt in self.SL_KEYS + self.ST_KEYS}
        slots, total = {}, 0
        for k, a in arrs.items():
            slots[k] = total
            total += (a.nbytes + 255) & ~255
        self.bytes = sum(a.nbytes for a in arrs.values())
        self.staging = torch.empty(total, dtype=torch.uint8).pin_memory()

        def views(buf):
            return {k: buf[slots[k]:slots[k] + a.nbytes].view(getattr(torch, a.dtype.name)).reshape(a.shape)
                    for k, a in arrs.items()}

        for k, t in views(self.staging).items():
            t.copy_(torch.from_numpy(arrs[k]))
        self.copy_stream = shared_stream("copy")
        self.sets = []
        for _ in range(2):
            buf = torch.empty(total, dtype=torch.uint8, device="cuda")
            d = views(buf)
            parts = []
            for k, (lo, hi) in enumerate(self.ranges):
                sb = mb.SLBatch(*(d[kk][lo:hi] for kk in ("s_layers", "l_row", "ego", "ob_s", "ob_l", "n_ob",
                                                            "ob_dist")), check=False)
                parts.append(dict(lo=lo, hi=hi, sb=sb, out=mb.SLPlanOut(hi - lo, N_S, N_L, want_maps=False), st=None,
                                  st_in=tuple(d[kk][lo:hi] for kk in ("st_n_s", "down", "up", "n_reg", "cur_vel")),
                                  stream=shared_stream("part%d" % k) if n_parts > 1 else None,
                                  done=torch.cuda.Event()))
            self.sets.append(dict(buf=buf, parts=parts, copied=torch.cuda.Event()))
        self.k = 0

    def step(self, _i=None):
        torch, mb = self.torch, self.mb
        s = self.sets[self.k & 1]
        with torch.cuda.stream(self.copy_stream):
            if self.k >= 2:
                for p in s["parts"]:
                    self.copy_stream.wait_event(p["done"])   # the kernels that read this buffer set two steps ago
            s["buf"].copy_(self.staging, non_blocking=True)
            s["copied"].record(self.copy_stream)
        for p in s["parts"]:
            stream = p["stream"] or torch.cuda.current_stream()
            stream.wait_event(s["copied"])
            with torch.cuda.stream(stream):
                mb.sl_plan(p["sb"], self.prm, p["out"])
                p["st"] = mb.st_dp(*p["st_in"], SPEED_LIM, want_tables=False, n_s_max=ST_N_S, out=p["st"])
                if self.rec_dev is not None:
                    k2 = self.k & 1
                    self.mdist.pack_results(p["out"].path_idx, p["out"].status, p["out"].layers_done,
                                            p["out"].best_cost, p["st"]["path_s_ind"], p["st"]["path_len"],
                                            out=self.rec_host[k2][p["lo"]:p["hi"]])   # stored over PCIe by the kernel
                p["done"].record(stream)
        if len(s["parts"]) == 1:
            p = s["parts"][0]
            self.gather.submit(p["out"].path_idx, p["out"].status, p["out"].layers_done, p["out"].best_cost,
                               p["st"]["path_s_ind"], p["st"]["path_len"])
        else:
            self.gather.submit_parts([(p["lo"], p["hi"], p["stream"],
                                       (p["out"].path_idx, p["out"].status, p["out"].layers_done, p["out"].best_cost,
                                        p["st"]["path_s_ind"], p["st"]["path_len"])) for p in s["parts"]])
        self.k += 1
        return s

    def cat(self, s, get):
        return self.torch.cat([get(p) for p in s["parts"]])


class SplitLoop:
    """The step with the batch cut into parts that never wait for one another: part k runs
    SL plan -> ST DP (-> its result records) on stream k.  Scenarios are independent, so
    this is the same work on the same kernels; what changes is that the tail of one part's
    kernel (the last workgroups, which leave most of the GPU idle) runs beside the other
    part's kernels, also across steps.  Only the result exchange looks at all parts."""

    def __init__(self, sc, prm, gather, n_parts, maps):
        import torch
        from motion_planning_amd import batch as mb, dist as mdist
        self.torch, self.mb, self.prm, self.gather, self.maps = torch, mb, prm, gather, maps
        B = sc["s_layers"].shape[0]
        self.parts = []
        for k in range(n_parts):
            lo, hi = mdist.shard_range(B, k, n_parts)
            sb = mb.SLBatch(sc["s_layers"][lo:hi], sc["l_row"][lo:hi], sc["ego"][lo:hi], sc["ob_s"][lo:hi],
                            sc["ob_l"][lo:hi], sc["n_ob"][lo:hi], sc["ob_dist"][lo:hi])
            st_in = (mb.to_dev(sc["st_n_s"][lo:hi], torch.int32), mb.to_dev(sc["down"][lo:hi]),
                     mb.to_dev(sc["up"][lo:hi]), mb.to_dev(sc["n_reg"][lo:hi], torch.int32),
                     mb.to_dev(sc["cur_vel"][lo:hi]))
            self.parts.append(dict(lo=lo, hi=hi, sb=sb, out=mb.SLPlanOut(hi - lo, N_S, N_L, want_maps=maps),
                                   st_in=st_in, st=None, stream=shared_stream("part%d" % k)))
        torch.cuda.synchronize()   # inputs were staged on the current stream

    def step(self, _i=None):
        torch, mb = self.torch, self.mb
        for p in self.parts:
            with torch.cuda.stream(p["stream"]):
                mb.sl_plan(p["sb"], self.prm, p["out"])
                p["st"] = mb.st_dp(*p["st_in"], SPEED_LIM, want_tables=self.maps, n_s_max=ST_N_S, out=p["st"])
        self.gather.submit_parts([(p["lo"], p["hi"], p["stream"],
                                   (p["out"].path_idx, p["out"].status, p["out"].layers_done, p["out"].best_cost,
                                    p["st"]["path_s_ind"], p["st"]["path_len"])) for p in self.parts])

    def cat(self, get):
        return self.torch.cat([get(p) for p in self.parts])


def bench_lattice(args, job):
    import torch
    from motion_planning_amd import batch as mb, dist as mdist

    world, rank, B = job.world, job.rank, job.B
    sc = make_scenarios(job.lo, B)
    sb = mb.SLBatch(sc["s_layers"], sc["l_row"], sc["ego"], sc["ob_s"], sc["ob_l"], sc["n_ob"], sc["ob_dist"])
    prm = mb.sl_params(early_exit=bool(args.early_exit), n_l=N_L)
    out = mb.SLPlanOut(B, N_S, N_L, want_maps=bool(args.maps))
    st_in = dict(n_s=mb.to_dev(sc["st_n_s"], torch.int32), s_down=mb.to_dev(sc["down"]), s_up=mb.to_dev(sc["up"]),
                 n_reg=mb.to_dev(sc["n_reg"], torch.int32), cur_vel=mb.to_dev(sc["cur_vel"]))
    gather = ResultGather(world, job.backend, B, job.rows, N_S, ST_N_T)
    gather_off = ResultGather(world, job.backend, B, job.rows, N_S, ST_N_T, active=False)
    steps, reps = args.steps, args.reps
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
            torch.cuda.Event(enable_timing=True)) for _ in range(steps)] for _ in range(reps)]
    state = dict(st=None, rep=-1, gather=gather)

    def step(i=None):
        e = ev[state["rep"]][i] if i is not None and state["rep"] >= 0 else None
        if e:
            e[0].record()
        mb.sl_plan(sb, prm, out)
        if e:
            e[1].record()
        state["st"] = mb.st_dp(st_in["n_s"], st_in["s_down"], st_in["s_up"], st_in["n_reg"], st_in["cur_vel"],
                               SPEED_LIM, want_tables=bool(args.maps), n_s_max=ST_N_S, out=state["st"])
        if e:
            e[2].record()
        st = state["st"]
        state["gather"].submit(out.path_idx, out.status, out.layers_done, out.best_cost, st["path_s_ind"],
                               st["path_len"])

    for _ in range(args.warmup):
        step()
    gather.drain()
    secs = []
    for r in range(reps):      # each block: barrier + sync, exactly `steps` steps, drain, barrier + sync
        state["rep"] = r
        secs += timed_blocks(job, step, gather.drain, steps, 1)
    state["rep"] = -1
    st = state["st"]

    sl_all = [e[0].elapsed_time(e[1]) for blk in ev for e in blk]
    st_all = [e[1].elapsed_time(e[2]) for blk in ev for e in blk]
    sl_ms, st_ms = float(np.mean(sl_all)), float(np.mean(st_all))
    layers = out.layers_done.cpu().numpy().astype(np.int64)
    edges_nominal = int(np.sum(N_L + (layers - 1) * N_L * N_L))
    found_frac = float((out.status == 1).float().mean())
    bad = int(((out.status != 1) & (out.status != 0)).sum())
    assert bad == 0, "kernel reported %d scenarios with an error status" % bad
    assert int(st["path_len"].min()) >= 2

    def check_exchange():
        if not job.dist_on:
            return
        # every rank holds every scenario's record, in scenario order
        u = mdist.unpack_results(gather.rows_of(rank, B).to("cuda"), N_S, ST_N_T)
        assert torch.equal(u["path_idx"], out.path_idx) and torch.equal(u["st_path"], st["path_s_ind"])
        assert torch.equal(u["status"], out.status) and torch.equal(u["best_cost"].isnan(), out.best_cost.isnan())
        n_all = 0
        for r in range(world):
            lo, hi = (mdist.shard_range(job.total, r, world) if job.scaling == "strong"
                      else (r * B, (r + 1) * B))
            n_all += int((mdist.unpack_results(gather.rows_of(r, hi - lo), N_S, ST_N_T)["st_len"] >= 2).sum())
        assert n_all == job.total, "gathered %d complete records of %d" % (n_all, job.total)

    check_exchange()

    # the same steps with the batch cut into parts on streams of their own (see SplitLoop)
    split_secs = None
    if args.streams > 1 and B >= 64 * args.streams:
        split = SplitLoop(sc, prm, gather, args.streams, bool(args.maps))
        for _ in range(max(2, args.warmup) + steps):   # (as many as a timed block: the first blocks of new streams are slow)
            split.step()
        gather.drain()
        split_secs = timed_blocks(job, split.step, gather.drain, steps, reps)
        torch.cuda.synchronize()
        assert torch.equal(split.cat(lambda p: p["out"].path_idx), out.path_idx)
        assert torch.equal(split.cat(lambda p: p["out"].status), out.status)
        assert torch.equal(split.cat(lambda p: p["st"]["path_s_ind"]), st["path_s_ind"])
        assert torch.equal(split.cat(lambda p: p["st"]["path_len"]), st["path_len"])
        check_exchange()

    # with an exchange in the step: the same steps of the headline schedule without it (same
    # kernels, same shard per rank), so that the record separates what the exchange costs from
    # what the batch size per GPU does
    noex_secs = None
    if job.dist_on:
        if split_secs is not None:
            split.gather = gather_off
            noex_secs = timed_blocks(job, split.step, gather_off.drain, steps, reps)
            split.gather = gather
        else:
            state["gather"] = gather_off
            noex_secs = timed_blocks(job, step, gather_off.drain, steps, reps)
            state["gather"] = gather
    who = job.describe()

    # the step again with the host->device copies of its inputs inside (never `value`)
    h2d = None
    if args.h2d_reps > 0:
        loop = H2DLoop(sc, prm, gather)
        for _ in range(max(2, args.warmup) + steps):   # the first copies of a run are slow
            loop.step()
        gather.drain()
        hsecs = timed_blocks(job, loop.step, gather.drain, steps, args.h2d_reps)
        last = loop.sets[(loop.k - 1) & 1]
        torch.cuda.synchronize()
        assert torch.equal(loop.cat(last, lambda p: p["out"].path_idx), out.path_idx)
        assert torch.equal(loop.cat(last, lambda p: p["st"]["path_s_ind"]), st["path_s_ind"])
        hsecs2 = None
        if split_secs:
            loop2 = H2DLoop(sc, prm, gather, args.streams)
            for _ in range(max(2, args.warmup) + steps):
                loop2.step()
            gather.drain()
            hsecs2 = timed_blocks(job, loop2.step, gather.drain, steps, args.h2d_reps)
            last = loop2.sets[(loop2.k - 1) & 1]
            torch.cuda.synchronize()
            assert torch.equal(loop2.cat(last, lambda p: p["out"].path_idx), out.path_idx)
            assert torch.equal(loop2.cat(last, lambda p: p["st"]["path_s_ind"]), st["path_s_ind"])
        # end to end: the headline schedule with the copy of the inputs in front and the results
        # handed over behind (exchange between the ranks, or records to pinned host memory)
        e2e_parts = args.streams if split_secs is not None else 1
        loop3 = H2DLoop(sc, prm, gather, e2e_parts, to_host=True)
        for _ in range(max(2, args.warmup) + steps):
            loop3.step()
        gather.drain()
        esecs = timed_blocks(job, loop3.step, gather.drain, steps, args.h2d_reps)
        torch.cuda.synchronize()
        if loop3.rec_host is not None:
            u = mdist.unpack_results(loop3.rec_host[(loop3.k - 1) & 1], N_S, ST_N_T)
            assert torch.equal(u["path_idx"], out.path_idx.cpu()) and torch.equal(u["st_path"], st["path_s_ind"].cpu())
            assert torch.equal(u["status"], out.status.cpu())
        else:
            check_exchange()
        h2d = (hsecs, loop.bytes, hsecs2, esecs, e2e_parts)

    # edges the fused kernel really prices (counting instance, outside the timed region)
    edges_eval = edges_serial = None
    if rank == 0 and not args.early_exit:
        cnt_out = mb.SLPlanOut(B, N_S, N_L, want_maps=False)
        _, edges_eval, edges_serial = mb.sl_plan_counted(sb, prm, cnt_out)
        assert torch.equal(cnt_out.path_idx, out.path_idx) and torch.equal(cnt_out.status, out.status)

    if rank != 0:
        return None
    # the headline is the schedule --streams names, whichever measured faster: two parts on two
    # streams by default (when the shard is large enough to cut), the whole batch on one stream
    # with --streams 1; both are always on the record
    one_sec = float(np.median(secs))
    one_secs = secs
    two_streams = split_secs is not None
    if two_streams:
        secs = split_secs
    sec = float(np.median(secs))
    plans_per_s = job.total * steps / sec

    def rate_stats(block_secs):
        if block_secs is None:
            return None
        return {"median": job.total * steps / float(np.median(block_secs)), "min": job.total * steps / max(block_secs),
                "max": job.total * steps / min(block_secs), "ms_per_step_median": 1e3 * float(np.median(block_secs)) / steps,
                "blocks": len(block_secs)}
    k_samples = 4  # samples per edge at a 1.5 m layer gap and 0.5 m spacing
    # algorithmic work per unit, as the reference writes it (1 flop per add/mul/sqrt/compare)
    flop_edge = 45 + k_samples * (50 + 7 * N_OB)      # one SL edge
    st_edges = B * (ST_N_S + (ST_N_T - 2) * ST_N_S * (ST_N_S + 1) // 2)  # ST edges per launch
    flop_st = float(30 + 9 * ST_R) * st_edges
    st_bytes = B * (16.0 * ST_N_T * ST_R + 4.0 * ST_N_T)  # interval bounds in, path out
    pmc = load_pmc()
    pmc_ok = pmc.get("_scenarios") == B    # the counters are per launch of the profiled batch size

    def traffic_of(kernel):
        k = pmc.get(kernel, {})
        return (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 if pmc_ok and "FETCH_SIZE" in k and "WRITE_SIZE" in k else None

    def issue_of(kernel, ms):
        k = pmc.get(kernel, {})
        return k["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e9 if pmc_ok and "SQ_INSTS_VALU" in k else None

    st_tf = flop_st / (st_ms * 1e-3) / 1e12
    st_issue = issue_of("st_dp_kernel", st_ms)
    st_name = ("st_dp_r_kernel<%d,4> (speed DP over the ST grid: the instance of st_dp_kernel for at most %d blocked "
               "intervals per layer, 4 lanes per cell)" % (ST_R, ST_R)) if B >= 2048 else \
        "st_dp_plain_kernel (speed DP over the ST grid: the kernel that holds every relaxation path)"
    # edges the speed DP prices, per plan, of the 42 874 of an 11 x 97 grid: counted by the counting
    # build of the same kernel (tools/count_st.py) on this batch; the product library carries no counters
    st_priced = ST_EDGES_PRICED_PER_PLAN * B if B >= 2048 else None
    roof_st = {"bound": "valu_issue", "kernel": st_name,
               "achieved": st_issue, "peak": ISSUE_PEAK_GINST, "unit": "G wave-instructions/s",
               "frac": st_issue / ISSUE_PEAK_GINST if st_issue else None,
               "traffic": traffic_of("st_dp_kernel"), "kernel_ms": st_ms,
               "algorithmic_flop_per_launch": flop_st, "st_edges_per_launch": st_edges,
               "edges_priced_per_launch": st_priced,
               "priced_fraction_of_grid": st_priced / st_edges if st_priced else None,
               "reference_work_tflops": st_tf, "reference_work_frac_of_fp64_peak": st_tf / VALU_FP64_PEAK_TF,
               "fp64_peak": VALU_FP64_PEAK_TF, "no_fma_ceiling": VALU_FP64_NO_FMA_TF,
               "hbm": {"achieved": st_bytes / (st_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": st_bytes / (st_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "algorithmic_bytes_per_launch": st_bytes},
               "note": "the kernel prices about a tenth of the grid's edges: exact pruning ends a cell's walk "
                       "early, and a first sweep over the cells a vehicle within the speed limit can reach bounds "
                       "the plan's cost so that nearly all other cells are never priced.  75 flop per ST edge "
                       "(30 + 9 R, R = 5) x every edge over the kernel time (reference_work_tflops) is therefore "
                       "what the reference would have had to do, not what the kernel does: it exceeds the 39.3 "
                       "TFLOP/s a kernel without fused multiply-adds can reach (results must equal Python's bit "
                       "for bit) and is not printed as a fraction.  frac = the kernel's own vector instructions "
                       "(SQ_INSTS_VALU of the committed counter pass) / duration against 1024 SIMDs x 2.4 GHz / "
                       "4 cycles; tables live in LDS, HBM traffic is the interval bounds in and the path out "
                       "(hbm: ~0 by design)"}
    sl_issue = issue_of("sl_plan_kernel", sl_ms)
    roof_sl = {"bound": "valu_issue", "kernel": "sl_plan_kernel<true,false> (fused edge cost + DP relaxation)",
               "achieved": sl_issue, "peak": ISSUE_PEAK_GINST, "unit": "G wave-instructions/s",
               "frac": sl_issue / ISSUE_PEAK_GINST if sl_issue else None,
               "traffic": traffic_of("sl_plan_kernel"), "kernel_ms": sl_ms,
               "edge_costs": "not materialised", "workgroups_per_cu": 8, "waves_per_simd": 4,
               "us_per_layer_per_launch": 1e3 * sl_ms / N_S,
               "edges_evaluated_per_launch": edges_eval,
               "lane_efficiency": edges_eval / (64.0 * edges_serial) if edges_serial else None,
               "edges_evaluated_per_sec": edges_eval / (sl_ms * 1e-3) if edges_eval else None,
               "evaluated_fraction_of_lattice": edges_eval / edges_nominal if edges_eval else None,
               "lattice_edges_nominal_per_launch": edges_nominal,
               "reference_flop_per_edge": flop_edge,
               "note": "the kernel keeps edge costs on chip, collapses the per-sample sums into per-layer "
                       "constants and prunes exactly, so neither 8 B nor 1645 flop per lattice edge describe its "
                       "work; it is bound by vector instruction issue and per-layer latency: frac = its own "
                       "vector instructions (SQ_INSTS_VALU of the committed counter pass) / duration against "
                       "1024 SIMDs x 2.4 GHz / 4 cycles.  The HBM-bound form of the same plan is "
                       "roofline_materialised."}
    dominant = roof_sl if sl_ms >= st_ms else roof_st
    res = {
        "metric": "SL+ST DP plans/sec (quintic lattice edges/sec: edges_evaluated_per_sec and "
                  "lattice_edges_nominal_per_sec)",
        "value": plans_per_s, "unit": "plans/s", "n_gpus": world, "steps": steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * sec / steps, "timed_reps": reps,
        "ms_per_step_min": 1e3 * min(secs) / steps, "ms_per_step_max": 1e3 * max(secs) / steps,
        "higher_is_better": True, "scaling": job.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "batched SL path DP %d scenarios over %d GPU(s) x (64 s-layers x 128 l-samples, "
                               "50 obstacles), every layer relaxed, + ST speed DP 11x97 with 5 blocked "
                               "intervals per layer; RCCL all-gather of results when n_gpus > 1"
                               % (job.total, world),
                   "total_scenarios": job.total, "scenarios_rank0": B, "per_gpu_batch": B,
                   "shards": [(lambda span: span[1] - span[0])(mdist.shard_range(job.total, r, world))
                              if job.scaling == "strong" else B for r in range(world)],
                   "n_s": N_S, "n_l": N_L,
                   "n_ob": N_OB,
                   "st_n_t": ST_N_T, "st_n_s": ST_N_S, "st_regions": ST_R, "early_exit": bool(args.early_exit),
                   "maps_written": bool(args.maps), "backend": job.backend if job.dist_on else None,
                   "parallelism": "scenario-sharded x%d" % world},
        "edges_evaluated_per_sec": world * edges_eval / (sl_ms * 1e-3) if edges_eval else None,
        "lattice_edges_nominal_per_sec": world * edges_nominal / (sl_ms * 1e-3),
        "schedule": ("the batch in %d parts on %d streams, each part SL plan -> ST DP on its own stream; parts "
                     "never wait for one another (scenarios are independent), so one part's kernels fill the "
                     "tail of the other's, also across steps" % (args.streams, args.streams)) if two_streams
                    else "one stream: SL plan -> ST DP over the whole batch",
        "schedule_chosen_by": "--streams %d (fixed by the flag, not by which schedule measured faster)" % args.streams,
        "value_one_stream": job.total * steps / one_sec,
        "value_two_streams": job.total * steps / float(np.median(split_secs)) if two_streams else None,
        "rates": {"one_stream": rate_stats(one_secs), "two_streams": rate_stats(split_secs)},
        "ms_per_step_one_stream": 1e3 * one_sec / steps,
        "ms_per_step_one_stream_min": 1e3 * min(one_secs) / steps,
        "kernel_times": "sl_kernel_ms / st_kernel_ms and every roofline: HIP events around each launch in the "
                        "one-stream timed blocks of this run (whole batch per launch, a kernel alone on the GPU)",
        "sl_kernel_ms": sl_ms, "st_kernel_ms": st_ms, "found_fraction": found_frac,
        "roofline": dominant,
        "roofline_other": roof_st if dominant is roof_sl else roof_sl,
        "pmc_source": pmc.get("_source"),
        "job": who,
    }
    if noex_secs is not None:
        # the same schedule, shard and kernels with the exchange switched off
        v_noex = job.total * steps / float(np.median(noex_secs))
        res["value_no_exchange"] = v_noex
        res["exchange_overhead_frac"] = 1.0 - plans_per_s / v_noex
        res["scenarios_rank0"] = B
        res["rates"]["no_exchange"] = rate_stats(noex_secs)
    if h2d:
        hsecs, nbytes, hsecs2, esecs, e2e_parts = h2d
        one_hsec = float(np.median(hsecs))
        hsec = float(np.median(hsecs2)) if (hsecs2 and two_streams) else one_hsec
        res["value_with_h2d"] = job.total * steps / hsec
        res["value_end_to_end"] = job.total * steps / float(np.median(esecs))
        res["h2d"] = {"ms_per_step": 1e3 * hsec / steps, "bytes_per_step_rank0": nbytes,
                      "copy_only_ms_at_63GBps": nbytes / 63e9 * 1e3,
                      "value_one_stream": job.total * steps / one_hsec,
                      "value_two_streams": job.total * steps / float(np.median(hsecs2)) if hsecs2 else None,
                      "rates": {"one_stream": rate_stats(hsecs), "two_streams": rate_stats(hsecs2),
                                "end_to_end": rate_stats(esecs)},
                      "ms_per_step_one_stream": 1e3 * one_hsec / steps,
                      "ms_per_step_parts": 1e3 * float(np.median(hsecs2)) / steps if hsecs2 else None,
                      "ms_per_step_parts_min": 1e3 * min(hsecs2) / steps if hsecs2 else None,
                      "ms_per_step_one_stream_min": 1e3 * min(hsecs) / steps,
                      "schedule": "one copy per step, then the kernels of two parts on two streams"
                                  if (hsecs2 and two_streams) else "one copy per step, then the kernels on one stream",
                      "end_to_end": {"ms_per_step": 1e3 * float(np.median(esecs)) / steps, "parts": e2e_parts,
                                     "results": "exchanged between the ranks (device-side packing + all-gather)"
                                                if gather.active else
                                                "packed on the device (one record per scenario) straight into pinned "
                                                "host memory, behind each part's kernels"},
                      "note": "the same steps with the inputs copied from one pinned staging buffer inside them: one "
                              "copy per step on a copy stream into one of two device buffers, so the copy of step "
                              "i+1 runs beside the kernels of step i; both schedules of the kernels are timed (the "
                              "parts wait for the copy, not for one another); value_with_h2d is the schedule "
                              "--streams names, value_end_to_end that schedule with the results handed over as "
                              "well (plans/s including the host-to-device copy and the result gather). "
                              "With two parts a timed block runs either at ms_per_step_parts_min (the copy hidden) "
                              "or about 0.2 ms slower (the copy exposed), block by block, so its median moves from "
                              "run to run; a copy per part instead of one per step ran slower than either (6.5 M)."}
    if args.materialised > 0 and world == 1:
        res["roofline_materialised"] = materialised_roofline(sc, min(args.materialised, B), prm)
    if args.cpu_sample > 0 and world == 1:  # the CPU leg belongs to the one-GPU run only
        threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        n = min(args.cpu_sample, B)
        v, dt, cres = cpu_baseline(sc, n, threads)
        res["cpu_baseline"] = {"value": v, "unit": "plans/s", "cores": threads, "kind": "port",
                               "sample": "first %d scenarios of the same batch (SL every layer + ST), "
                                         "C oracle, %d threads, %.1f s" % (n, threads, dt),
                               "reference_python": {"value": 1.0 / 1500.0, "unit": "plans/s", "cores": 1,
                                                    "note": "the reference itself, timed in the build container "
                                                            "only (its source does not travel): ~368 us per SL "
                                                            "edge -> ~25 min per 64x128x50 plan, plus 0.4 s per "
                                                            "ST plan of 101 cells"}}
        if not args.early_exit and not args.maps:
            ties = check_lattice_parity(cres, out, st, sc)
            res["cpu_baseline"]["parity_checked"] = n
            res["cpu_baseline"]["parity_ties"] = ties
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=15, help="timed blocks of --steps steps; the median is reported")
    ap.add_argument("--scenarios", type=int, default=None, help="scenarios per GPU (weak scaling)")
    ap.add_argument("--total-scenarios", type=int, default=None,
                    help="scenarios split over all GPUs (strong scaling; default 65536 when --gpus > 1)")
    ap.add_argument("--workload", choices=["lattice", "pipeline"], default="lattice",
                    help="lattice: batched SL path DP (64x128 lattice, 50 obstacles) + ST speed DP; "
                         "pipeline: the whole replan on curved reference lines with 256 obstacles")
    ap.add_argument("--cpu-sample", type=int, default=448, help="scenarios timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--early-exit", type=int, default=0, help="1 = stop a scenario at its first failing layer")
    ap.add_argument("--materialised", type=int, default=1024,
                    help="scenarios run through the two-kernel SL path for its HBM roofline (0 = skip)")
    ap.add_argument("--maps", type=int, default=0, help="1 = also write cost_map/index_map/ST tables to HBM")
    ap.add_argument("--streams", type=int, default=2,
                    help="parts the batch is cut into for the headline step, each on a stream of its own "
                         "(1 = the whole batch on one stream)")
    ap.add_argument("--h2d-reps", type=int, default=9, help="timed blocks of the host-to-device variant (0 = skip)")
    ap.add_argument("--pipeline-steps", type=int, default=10,
                    help="steps per timed block of the whole-replan leg attached to the lattice line (0 = skip)")
    ap.add_argument("--pipeline-scenes", type=int, default=4096, help="scenes per GPU of the whole-replan workload")
    ap.add_argument("--pipeline-cpu-sample", type=int, default=256)
    ap.add_argument("--single-plan", type=int, default=200,
                    help="calls per single-plan latency measurement (one plan per call; 0 = skip)")
    ap.add_argument("--launch-check", action="store_true",
                    help="exercise the launcher, the sharding and the result exchange only (CPU, gloo; no kernels)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    job = Job(args)
    if args.launch_check:
        launch_check(job)
        job.close()
        return
    if args.workload == "pipeline":
        res = bench_pipeline(args, job, args.steps, args.warmup, max(1, args.reps), args.cpu_sample)
    else:
        res = bench_lattice(args, job)
        if args.pipeline_steps > 0:
            pipe = bench_pipeline(args, job, args.pipeline_steps, 2, 5, args.pipeline_cpu_sample)
            if res is not None:
                res["pipeline"] = pipe
        if args.single_plan > 0 and job.world == 1 and res is not None:
            res["single_plan"] = single_plan_latencies(args.single_plan)
    if job.rank == 0:
        print(json.dumps(res))
    job.close()


if __name__ == "__main__":
    main()
