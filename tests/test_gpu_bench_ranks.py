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
    assert d["value"] >= d["value_one_stream"] > 0
    assert d["pipeline"]["n_gpus"] == 2 and d["pipeline"]["value"] >= d["pipeline"]["value_one_stream"] > 0
