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


@pytest.mark.gpu
def test_two_ranks_with_the_hip_engine_and_a_collective_on_one_gpu():
    """The N > 1 bench path with real engines: `bench.py --gpus 2` spawns two ranks, each runs its own chain through libertirt.so on cuda:0
    (ERM_BENCH_REHEARSE=1: a one-GPU box refuses two RCCL ranks on one device, so the collectives go over gloo), the ranks barrier around the
    timed region, MAX-reduce their times and all-reduce the posterior summaries."""
    e = dict(os.environ, ERM_BENCH_REHEARSE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "ERM_BENCH_LAUNCH_ONLY"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4", "--nsubj", "4000", "--nitem", "12",
                        "--clock-warmup-ms", "0"], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert out["n_gpus"] == 2 and out["config"]["chains"] == 2 and out["config"]["collective_backend"] == "gloo" and out["scaling"] == "weak"
    assert out["dtype"] == "f64" and out["fp32"]["dtype"] == "f32" and out["gather_ms"] > 0
    assert out["value"] == pytest.approx(4000 * 12 * 12 * 2 / (out["ms_per_step"] * 1e-3 * 12), rel=1e-6)      # whole-job aggregate over both ranks
    assert out["roofline"]["launches_timed"] >= 8 and out["roofline"]["algorithmic_bytes_per_cell_update"] == 25
    c4 = out["configs4"]                          # configs[4]'s per-GPU load (rehearsal size), data generated on the device
    assert c4["value"] > 0 and "nItem=100" in c4["workload"] and "device" in c4["data"]
