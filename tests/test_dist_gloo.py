"""Scenario sharding + result gather with world_size 2 over gloo on the CPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from motion_planning_amd import dist as mdist

N_S, N_T = 6, 11


def fake_records(lo, hi):
    """Deterministic per-scenario results as the kernels would produce them."""
    ids = torch.arange(lo, hi, dtype=torch.int32)
    B = hi - lo
    path_idx = (ids[:, None] * 7 + torch.arange(N_S, dtype=torch.int32)[None]) % 5
    status = (ids % 3 != 0).to(torch.int32)
    layers = torch.where(status == 1, torch.full((B,), N_S, dtype=torch.int32), ids % N_S + 1)
    best = torch.where(status == 1, ids.double() * 1.25 + 0.1, torch.full((B,), float("nan"), dtype=torch.float64))
    st_len = (ids % N_T + 1).to(torch.int32)
    st_path = torch.where(torch.arange(N_T)[None] < st_len[:, None],
                          (ids[:, None] + torch.arange(N_T, dtype=torch.int32)[None]) % 29,
                          torch.full((B, N_T), -1, dtype=torch.int32)).to(torch.int32)
    return mdist.pack_results(path_idx, status, layers, best, st_path, st_len)


def worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = mdist.shard_range(n_total, rank, world)
    got = mdist.gather_results(fake_records(lo, hi))
    want = fake_records(0, n_total)
    ok = torch.equal(got, want)
    u = mdist.unpack_results(got, N_S, N_T)
    ok = ok and u["best_cost"].shape == (n_total,) and bool(torch.isnan(u["best_cost"][0]))
    ok = ok and float(u["best_cost"][1]) == 1.35 and u["st_path"].shape == (n_total, N_T)
    q.put((rank, ok, tuple(got.shape)))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_total", [64, 37])   # equal shards, ragged shards
def test_gather_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, "rank %d gathered wrong records" % rank
        assert shape == (n_total, mdist.record_width(N_S, N_T))


@pytest.mark.parametrize("flags,total,shards", [(["--total-scenarios", "37"], 37, [19, 18]),
                                                (["--scenarios", "12"], 24, [12, 12]),
                                                ([], 65536, [32768, 32768])])
def test_bench_launches_its_own_ranks(flags, total, shards):
    """`python bench.py --gpus 2` with no launcher in the environment starts two ranks by
    itself; --launch-check runs the sharding and the padded result exchange of the
    benchmark over gloo on CPU tensors (no kernels), and rank 0 prints one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"] + flags,
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["launch_check"] == "ok" and r["n_gpus"] == 2 and r["total_scenarios"] == total
    assert r["shards"] == shards and r["backend"] == "gloo"
    assert r["scaling"] == ("weak" if "--scenarios" in flags else "strong")
    # what a multi-rank line says about the exchange and about who took part
    assert r["value"] > 0 and r["value_no_exchange"] > 0 and r["exchange_overhead_frac"] < 1.0
    assert r["scenarios_rank0"] == shards[0] == r["config"]["per_gpu_batch"]
    job = r["job"]
    assert job["world_size"] == 2 and job["backend"] == "gloo"
    assert [q["rank"] for q in job["ranks"]] == [0, 1] and len({q["pid"] for q in job["ranks"]}) == 2


def test_shard_range_covers_everything():
    for n, w in [(4096, 8), (65536, 8), (37, 2), (5, 8), (0, 4)]:
        spans = [mdist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip_single_process():
    rec = fake_records(0, 10)
    u = mdist.unpack_results(rec, N_S, N_T)
    again = mdist.pack_results(u["path_idx"], u["status"], u["layers_done"], u["best_cost"], u["st_path"],
                               u["st_len"])
    assert torch.equal(rec, again)
    assert mdist.gather_results(rec) is rec   # world size 1: nothing to do
    assert np.isnan(float(u["best_cost"][0]))
