"""bench.py with two ranks on the one GPU of the test box (gloo carries the exchange, the
ranks share the card): the self-launch, both schedules of the step, the per-part packing of
the result records and bench.py's own comparisons of parts, ranks and exchange all run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_both_schedules_and_the_exchange():
    env = dict(os.environ, MP_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scenarios", "512", "--steps", "2",
           "--reps", "2", "--warmup", "1", "--cpu-sample", "0", "--materialised", "0", "--h2d-reps", "1",
           "--pipeline-steps", "2", "--pipeline-scenes", "256"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["total_scenarios"] == 1024 and d["scaling"] == "weak"
    assert d["config"]["backend"] == "gloo"
    # both schedules were timed (the parts' results and the gathered records are compared
    # with the one-stream results inside bench.py; a mismatch is an AssertionError there)
    assert d["value"] == d["value_two_streams"] > 0 and d["value_one_stream"] > 0    # --streams 2 is the default
    assert d["pipeline"]["n_gpus"] == 2
    assert d["pipeline"]["value"] == d["pipeline"]["value_two_streams"] > 0 and d["pipeline"]["value_one_stream"] > 0
    # the exchange priced on its own, the end-to-end rate, and who took part
    assert d["value_no_exchange"] > 0 and d["exchange_overhead_frac"] < 1.0 and d["scenarios_rank0"] == 512
    assert d["value_end_to_end"] > 0 and d["value_with_h2d"] > 0
    assert d["job"]["world_size"] == 2 and [q["rank"] for q in d["job"]["ranks"]] == [0, 1]
    assert all("pci_bus_id" in q for q in d["job"]["ranks"])


def test_two_ranks_split_the_65536_scenarios():
    """The default of `bench.py --gpus 2`: the 65536-scenario configuration cut into two
    contiguous shards of 32768 (strong scaling), every step's records exchanged and checked."""
    env = dict(os.environ, MP_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--reps", "2", "--warmup", "1",
           "--cpu-sample", "0", "--materialised", "0", "--h2d-reps", "0", "--pipeline-steps", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["total_scenarios"] == 65536 and d["config"]["shards"] == [32768, 32768]
    assert d["config"]["per_gpu_batch"] == 32768 and d["scenarios_rank0"] == 32768
    assert d["value"] > 0 and d["value_no_exchange"] > 0 and d["job"]["world_size"] == 2


def test_one_rank_exchange_over_rccl():
    """The exchange as the multi-GPU run issues it (device-side record packing, asynchronous
    all_gather_into_tensor on device tensors through the "nccl" backend, which is RCCL here), with
    a group of one rank: MP_BENCH_FORCE_DIST=1 switches it on for a single process."""
    env = dict(os.environ, MP_BENCH_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MP_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scenarios", "512", "--steps", "2",
           "--reps", "2", "--warmup", "1", "--cpu-sample", "0", "--materialised", "0", "--h2d-reps", "1",
           "--pipeline-steps", "2", "--pipeline-scenes", "256", "--single-plan", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl"
    assert d["job"]["backend"] == "nccl" and "RCCL" in d["job"]["collective_library"]
    assert d["value"] > 0 and d["value_no_exchange"] > 0 and d["exchange_overhead_frac"] < 1.0
    assert d["value_end_to_end"] > 0 and d["pipeline"]["value"] > 0
