#!/usr/bin/env python
"""Benchmark: batched SL path DP + ST speed DP plans/sec on MI355X.

One step = one pass of the hot path over one batch of synthetic scenarios that
already sit in HBM: the SL path DP over a 64-layer x 128-lane lattice with 50
obstacles per scenario (every layer relaxed, no early abort) followed by the
ST speed DP on a default ST lattice (11 time layers x 97 s cells, 5 moving
blocked intervals per layer), and -- with more than one GPU -- one RCCL
all-gather of the per-scenario results.  Scenarios are independent, so ranks
take disjoint blocks of scenarios (weak scaling: 4096 per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]

prints ONE JSON line (rank 0).  For N > 1 launch with torch.distributed.run.
"""
import argparse
import json
import os
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
VALU_FP32_PEAK_TF = 157.3    # MI355X vector fp32 spec
VALU_FP64_PEAK_TF = 78.6     # MI355X vector fp64 spec


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
                dy_rad=np.full((B, n_dy), 0.8), n_dy=np.full(B, n_dy, np.int32))


def cpu_pipeline(sc, n_sample, threads):
    """The CPU oracle's whole replan (Frenet map, SL DP, path sampling, ST map, ST DP,
    speed profile, conversion back) on the first n_sample scenes, `threads` at a time."""
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
            return 0
        found, curve, _ = orc.sl_plan(sl)
        if not found:
            return 0
        pb = sl.path_sampling(curve)
        n_s, n_t = orc.st_dims(pb[-1][0])
        down, up, n_reg = orc.st_add_obstacle(n_t, pb, sl.st_ob_traj, sl.dy_ob_traj)
        sr = orc.st_find_speed(n_t, n_s, down, up, n_reg, sl.robot_vel, SPEED_LIM)
        ss, vv = orc.speed_profile(sr["path_s_ind"], sl.robot_vel)
        sl.path_convert(sl.path_sampling(curve, ss))
        return 1

    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()), ThreadPoolExecutor(threads) as ex:
        list(ex.map(one, range(n_sample)))
    dt = time.perf_counter() - t0
    return n_sample / dt, dt


def exchange_on(world):
    """The result exchange runs with more than one rank; MP_BENCH_FORCE_DIST=1 also runs it
    (process group, barrier, all-gather, max over ranks) in a one-rank job, which is how the
    RCCL calls are rehearsed on a one-GPU box."""
    return world > 1 or os.environ.get("MP_BENCH_FORCE_DIST") == "1"


class ResultGather:
    """The per-step exchange of result records when more than one GPU takes part: pack on
    the device (one kernel), then one all-gather issued asynchronously on the process
    group's own stream into one of two buffer pairs, so that the exchange of batch k runs
    while the kernels of batch k+1 do; a buffer pair is waited for before it is reused,
    and drain() waits for everything (called before the clock stops)."""

    def __init__(self, world, backend, B, n_s, n_t):
        import torch
        from motion_planning_amd import dist as mdist
        self.world, self.backend, self.mdist = world, backend, mdist
        self.k = 0
        self.active = exchange_on(world)
        if not self.active:
            return
        width = mdist.record_width(n_s, n_t)
        gdev = "cuda" if backend == "nccl" else "cpu"
        self.packed = [torch.empty((B, width), dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.gathered = [torch.empty((world * B, width), dtype=torch.uint8, device=gdev) for _ in range(2)]
        self.pending = [None, None]

    def submit(self, path_idx, status, layers_done, best_cost, st_path, st_len):
        if not self.active:
            return
        i = self.k & 1
        if self.pending[i] is not None:
            self.pending[i].wait()
        self.mdist.pack_results(path_idx, status, layers_done, best_cost, st_path, st_len, out=self.packed[i])
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


def bench_pipeline(args, world, rank, backend):
    """--workload pipeline: one step = one mp_plan_scenes call over B scenes resident in
    HBM (every stage of the replan) and, with more than one GPU, the result gather."""
    import torch
    import torch.distributed as dist
    from motion_planning_amd import batch as mb, dist as mdist, pipeline as pl

    B = args.scenarios
    sc = make_scenes(B, seed=5 + rank)
    dev = {k: mb.to_dev(v, torch.int32 if v.dtype == np.int32 else torch.float64) for k, v in sc.items()}
    dev["line_vec"] = torch.zeros(B, dtype=torch.float64, device="cuda")
    params = pl.default_params()
    dims = pl.plan_dims(B, sc["knots_x"].shape[1], sc["st_pos"].shape[1], sc["dy_pos"].shape[1], n_l=5, m_max=200,
                        params=params)
    arena = pl.PlanArena(dims)
    a = arena.arrays
    gather = ResultGather(world, backend, B, dims.n_s_max, dims.n_t)

    def step():
        pl.plan_scenes(dev, dims, params, arena)
        gather.submit(a["path_idx"], a["sl_status"], a["layers_done"], a["best_cost"], a["st_path_s_ind"],
                      a["st_path_len"])

    def fence():
        torch.cuda.synchronize()
        if exchange_on(world):
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gather.drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gather.drain()
    fence()
    elapsed = time.perf_counter() - t0
    if exchange_on(world):
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        u = mdist.unpack_results(gather.last()[rank * B:(rank + 1) * B].to("cuda"), dims.n_s_max, dims.n_t)
        assert torch.equal(u["path_idx"], a["path_idx"]) and torch.equal(u["st_path"], a["st_path_s_ind"])
    status = a["status"].cpu().numpy()
    assert set(np.unique(status)) <= {1, 0, -4}, "unexpected scene status %s" % np.unique(status)
    n_prof = a["n_prof"].cpu().numpy()
    assert (n_prof[status == 1] >= 2).all()
    if rank != 0:
        return
    n_pts = B * (sc["st_pos"].shape[1] + sc["dy_pos"].shape[1] * 11 + 1)
    bytes_in = sum(v.nbytes for v in sc.values())
    bytes_out = B * (8 * 3 * dims.q_max + 4 * 2)   # final x, y, speed profile, status, count
    ms = 1e3 * elapsed / args.steps
    res = {
        "metric": "full replan plans/sec (Cartesian->Frenet + SL path DP + ST speed DP + speed profile + path in "
                  "the robot frame)",
        "value": world * B * args.steps / elapsed, "unit": "plans/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "full pipeline: %d scenes/GPU, curved reference lines of 16-21 knots, 192 static + "
                               "64 moving obstacles (897 points to Frenet per scene), default lattice (5 lanes, "
                               "layers every 8 m), default ST grid; one mp_plan_scenes call per step" % B,
                   "scenarios_per_gpu": B, "n_l": 5, "n_st": 192, "n_dy": 64,
                   "parallelism": "scenario-sharded x%d" % world},
        "c2f_points_per_sec": world * n_pts * args.steps / elapsed,
        "status_fractions": {"planned": float((status == 1).mean()), "no_path": float((status == 0).mean()),
                             "map_longer_than_100m": float((status == -4).mean())},
        "roofline": {"bound": "hbm", "kernel": "whole chain (13 kernels, see profiles/ for the split)",
                     "achieved": (bytes_in + bytes_out) / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (bytes_in + bytes_out) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": bytes_in + bytes_out,
                     "note": "algorithmic bytes = the scenes in and the plans out; every stage in between works "
                             "from LDS-resident tables, so the chain is bound by instruction issue and latency, "
                             "not by HBM"},
    }
    if args.cpu_sample > 0 and world == 1:  # the CPU leg belongs to the one-GPU run only
        threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        n = min(args.cpu_sample, B)
        v, dt = cpu_pipeline(sc, n, threads)
        res["cpu_baseline"] = {"value": v, "unit": "plans/s", "cores": threads, "kind": "port",
                               "sample": "first %d scenes of the same batch, whole replan, C oracle driven from "
                                         "Python, %d threads, %.1f s" % (n, threads, dt)}
    print(json.dumps(res))


def load_pmc():
    """Per-kernel counter means (per launch) from the last committed rocprofv3 --pmc passes of this
    same command; bench.py cannot collect them itself."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        k = dict(d.get("kernels", {}))
        k["_source"] = d.get("source")
        return k
    except (OSError, ValueError):
        return {}


def cpu_baseline(sc, n_sample, threads):
    """The CPU oracle (a quirk-exact C restatement of the reference planner) on the
    first n_sample scenarios of the same batch, `threads` scenarios at a time."""
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
        orc.st_find_speed(ST_N_T, ST_N_S, sc["down"][b], sc["up"][b], sc["n_reg"][b], float(sc["cur_vel"][b]),
                          SPEED_LIM)
        return r["found"]

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(one, range(n_sample)))
    dt = time.perf_counter() - t0
    return n_sample / dt, dt


def materialised_roofline(sc, n, prm):
    """The two-kernel form of the same SL plan on the first n scenarios, outside the timed
    region: mp_sl_edge_cost writes every edge cost to HBM (fp64, 8 B/edge) and mp_sl_dp
    streams that tensor once -- the HBM-bound relaxation the fused kernel makes unnecessary,
    measured so that an HBM roofline fraction below 1 is on record."""
    import torch
    from motion_planning_amd import batch as mb
    sb = mb.SLBatch(sc["s_layers"][:n], sc["l_row"][:n], sc["ego"][:n], sc["ob_s"][:n], sc["ob_l"][:n],
                    sc["n_ob"][:n], sc["ob_dist"][:n])
    edge, _ = mb.sl_edge_cost(sb, prm)
    out = mb.sl_dp(edge, prm)
    torch.cuda.synchronize()
    # (the tests compare this path with the fused kernel bit for bit; launching the fused
    # kernel here would mix a launch of another size into its per-kernel profile)
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
    return {"bound": "hbm", "kernel": "sl_dp_kernel (relaxation over the materialised fp64 edge tensor)",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "kernel_ms": t_d, "algorithmic_bytes_per_launch": nbytes, "scenarios": n,
            "edge_kernel": {"kernel": "sl_edge_cost_kernel", "kernel_ms": t_e,
                            "written_GBps": nbytes / (t_e * 1e-3) / 1e9,
                            "edges_per_sec": n * N_S * N_L * N_L / (t_e * 1e-3)},
            "note": "not part of the timed step: the product path is the fused kernel (roofline / roofline_other); "
                    "this is the same plan through mp_sl_edge_cost + mp_sl_dp, whose DP kernel reads each edge "
                    "cost once from HBM"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenarios", type=int, default=4096, help="scenarios per GPU")
    ap.add_argument("--workload", choices=["lattice", "pipeline"], default="lattice",
                    help="lattice: batched SL path DP (64x128 lattice, 50 obstacles) + ST speed DP; "
                         "pipeline: the whole replan on curved reference lines with 256 obstacles")
    ap.add_argument("--cpu-sample", type=int, default=448, help="scenarios timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--early-exit", type=int, default=0, help="1 = stop a scenario at its first failing layer")
    ap.add_argument("--materialised", type=int, default=1024,
                    help="scenarios run through the two-kernel SL path for its HBM roofline (0 = skip)")
    ap.add_argument("--maps", type=int, default=0, help="1 = also write cost_map/index_map/ST tables to HBM")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from motion_planning_amd import batch as mb, dist as mdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # MP_BENCH_BACKEND=gloo rehearses the multi-rank path on a one-GPU box (ranks
    # share the card and the gather is staged through host memory)
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if exchange_on(world):
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)   # a forced one-rank job started without a launcher
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    if args.workload == "pipeline":
        bench_pipeline(args, world, rank, backend)
        if exchange_on(world):
            dist.destroy_process_group()
        return
    B = args.scenarios
    sc = make_scenarios(rank * B, B)
    sb = mb.SLBatch(sc["s_layers"], sc["l_row"], sc["ego"], sc["ob_s"], sc["ob_l"], sc["n_ob"], sc["ob_dist"])
    prm = mb.sl_params(early_exit=bool(args.early_exit), n_l=N_L)
    out = mb.SLPlanOut(B, N_S, N_L, want_maps=bool(args.maps))
    st_in = dict(n_s=mb.to_dev(sc["st_n_s"], torch.int32), s_down=mb.to_dev(sc["down"]), s_up=mb.to_dev(sc["up"]),
                 n_reg=mb.to_dev(sc["n_reg"], torch.int32), cur_vel=mb.to_dev(sc["cur_vel"]))
    gather = ResultGather(world, backend, B, N_S, ST_N_T)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        mb.sl_plan(sb, prm, out)
        if i is not None:
            ev[i][1].record()
        st = mb.st_dp(st_in["n_s"], st_in["s_down"], st_in["s_up"], st_in["n_reg"], st_in["cur_vel"], SPEED_LIM,
                      want_tables=bool(args.maps), n_s_max=ST_N_S)
        if i is not None:
            ev[i][2].record()
        gather.submit(out.path_idx, out.status, out.layers_done, out.best_cost, st["path_s_ind"], st["path_len"])
        return st

    def fence():
        torch.cuda.synchronize()
        if exchange_on(world):
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gather.drain()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        st = step(i)
    gather.drain()
    fence()
    elapsed = time.perf_counter() - t0
    if exchange_on(world):
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    sl_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    st_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    layers = out.layers_done.cpu().numpy().astype(np.int64)
    edges_launch = int(np.sum(N_L + (layers - 1) * N_L * N_L))
    found_frac = float((out.status == 1).float().mean())
    bad = int(((out.status != 1) & (out.status != 0)).sum())
    assert bad == 0, "kernel reported %d scenarios with an error status" % bad
    assert int(st["path_len"].min()) >= 2

    if exchange_on(world):
        # every rank holds every scenario's record, in scenario order
        gathered = gather.last()
        u = mdist.unpack_results(gathered[rank * B:(rank + 1) * B].to("cuda"), N_S, ST_N_T)
        assert torch.equal(u["path_idx"], out.path_idx) and torch.equal(u["st_path"], st["path_s_ind"])
        assert torch.equal(u["status"], out.status) and torch.equal(u["best_cost"].isnan(), out.best_cost.isnan())
        assert int((mdist.unpack_results(gathered, N_S, ST_N_T)["st_len"] >= 2).sum()) == world * B
    if rank == 0:
        plans_per_s = world * B * args.steps / elapsed
        edges_per_s = edges_launch / (sl_ms * 1e-3)
        k_samples = 4  # samples per edge at a 1.5 m layer gap and 0.5 m spacing
        # algorithmic work per unit, as the reference writes it (1 flop per add/mul/sqrt/compare)
        flop_edge = 45 + k_samples * (50 + 7 * N_OB)      # one SL edge
        sl_bytes = 8.0 * edges_launch                     # 4 B written + 4 B read per edge if materialised
        st_edges = B * (ST_N_S + (ST_N_T - 2) * ST_N_S * (ST_N_S + 1) // 2)  # ST edges per launch
        flop_st_edge = 30 + 9 * ST_R                      # one ST edge
        st_bytes = B * (16.0 * ST_N_T * ST_R + 4.0 * ST_N_T)  # interval bounds in, path out
        pmc = load_pmc()

        def roof(kernel, label, ms, alg_bytes, alg_flop, note):
            k = pmc.get(kernel, {})
            traffic = (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in k and "WRITE_SIZE" in k else None
            gbs = alg_bytes / (ms * 1e-3) / 1e9
            tf = alg_flop / (ms * 1e-3) / 1e12
            r = {"bound": "hbm", "kernel": label, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": ms,
                 "algorithmic_bytes_per_launch": alg_bytes,
                 "valu": {"achieved": tf, "peak": VALU_FP64_PEAK_TF, "unit": "TFLOP/s (fp64 vector)",
                          "frac": tf / VALU_FP64_PEAK_TF, "algorithmic_flop_per_launch": alg_flop},
                 "note": note}
            if "SQ_INSTS_VALU" in k:
                # share of the chip's vector issue slots the kernel's own instructions filled:
                # 4 cycles per wave64 instruction, 1024 SIMDs, nominal 2.4 GHz
                r["valu"]["issue_slots_used"] = k["SQ_INSTS_VALU"] * 4.0 / (1024 * ms * 1e-3 * 2.4e9)
            return r

        roof_sl = roof("sl_plan_kernel", "sl_plan_kernel (fused edge cost + DP relaxation)", sl_ms, sl_bytes,
                       float(flop_edge) * edges_launch,
                       "algorithmic bytes = 8 B/edge (edge cost written once and read once if materialised) and "
                       "1645 flop/edge as the reference writes them; the fused kernel keeps edge costs on chip, "
                       "collapses the per-sample sums into per-layer constants and prunes exactly, so both "
                       "fractions exceed 1: the kernel is bound by instruction latency per layer, not by HBM or "
                       "flops (traffic = measured FETCH_SIZE+WRITE_SIZE per launch, 8-byte-per-lane accesses, "
                       "uncorrected)")
        roof_st = roof("st_dp_kernel", "st_dp_kernel (speed DP over the ST grid)", st_ms, st_bytes,
                       float(flop_st_edge) * st_edges,
                       "tables live in LDS: HBM traffic is the interval bounds in and the path out, so the HBM "
                       "fraction is ~0 by design; the bound that matters is the fp64 vector rate (valu)")
        dominant = roof_sl if sl_ms >= st_ms else roof_st
        res = {
            "metric": "SL+ST DP plans/sec (quintic lattice edges/sec in edges_per_sec)",
            "value": plans_per_s, "unit": "plans/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "batched SL path DP 4096 scenarios/GPU x (64 s-layers x 128 l-samples, "
                                   "50 obstacles), every layer relaxed, + ST speed DP 11x97 with 5 blocked "
                                   "intervals per layer; RCCL all-gather of results when n_gpus > 1",
                       "scenarios_per_gpu": B, "n_s": N_S, "n_l": N_L, "n_ob": N_OB, "st_n_t": ST_N_T,
                       "st_n_s": ST_N_S, "st_regions": ST_R, "early_exit": bool(args.early_exit),
                       "maps_written": bool(args.maps), "parallelism": "scenario-sharded x%d" % world},
            "edges_per_sec": edges_per_s * world, "edges_per_launch": edges_launch,
            "sl_kernel_ms": sl_ms, "st_kernel_ms": st_ms, "found_fraction": found_frac,
            "roofline": dominant,
            "roofline_other": roof_st if dominant is roof_sl else roof_sl,
            "pmc_source": pmc.get("_source"),
        }
        if args.materialised > 0 and world == 1:
            res["roofline_materialised"] = materialised_roofline(sc, min(args.materialised, B), prm)
        if args.cpu_sample > 0 and world == 1:  # the CPU leg belongs to the one-GPU run only
            threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            v, dt = cpu_baseline(sc, min(args.cpu_sample, B), threads)
            res["cpu_baseline"] = {"value": v, "unit": "plans/s", "cores": threads, "kind": "port",
                                   "sample": "first %d scenarios of the same batch (SL every layer + ST), "
                                             "C oracle, %d threads, %.1f s" % (min(args.cpu_sample, B), threads, dt)}
        print(json.dumps(res))
    if exchange_on(world):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
