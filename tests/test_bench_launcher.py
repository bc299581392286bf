"""bench.py --gpus N must create its own N ranks when it is not running under torch.distributed.run (the driver's N=1-shaped command
line with a larger N).  CPU test of the launcher: ERM_BENCH_LAUNCH_ONLY=1 stops each rank after the rendezvous and one all-reduce
(gloo), before any engine or GPU call; the parent relays rank 0's JSON line and the children's exit codes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, ERM_BENCH_REHEARSE="1", ERM_BENCH_LAUNCH_ONLY="1", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_flag_spawns_that_many_ranks(n):
    r = _run(["--gpus", str(n), "--steps", "7"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # exactly one line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["ranks"] == n and out["backend"] == "gloo" and out["steps"] == 7
    assert out["rank_sum"] == n * (n + 1) / 2               # every rank took part in the collective


def test_single_gpu_invocation_spawns_nothing():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["backend"] is None


def test_world_size_mismatch_is_an_error():
    e = dict(os.environ, ERM_BENCH_REHEARSE="1", ERM_BENCH_LAUNCH_ONLY="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
