"""bench.py --gpus N.  Default at N > 1: the library's chain farm driven by ONE process (rank 0 under torch.distributed.run, whose other ranks only
join the barriers over gloo).  `--multiprocess` keeps one process per GPU; without torch.distributed.run it creates its own N ranks.  CPU tests
of both launch shapes: ERM_BENCH_LAUNCH_ONLY=1 stops each rank after the rendezvous and one all-reduce (gloo), before any engine or GPU call."""
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
    r = _run(["--gpus", str(n), "--steps", "7", "--multiprocess"])
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


@pytest.mark.parametrize("extra", [[], ["--multiprocess"]])
def test_world_size_mismatch_is_an_error(extra):
    e = dict(os.environ, ERM_BENCH_REHEARSE="1", ERM_BENCH_LAUNCH_ONLY="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"] + extra, env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_farm_mode_is_one_process_and_joins_a_torchrun_launch():
    """`python bench.py --gpus 3` (no torch.distributed.run): the farm needs no ranks, nothing is spawned.  Under the driver's launch shape
    (`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`) both ranks rendezvous over gloo and rank 0 prints the one line."""
    r = _run(["--gpus", "3", "--steps", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["mode"] == "farm" and out["n_gpus"] == 3 and out["ranks"] == 1 and out["backend"] is None
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        e = dict(os.environ, ERM_BENCH_REHEARSE="1", ERM_BENCH_LAUNCH_ONLY="1", WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    lines = [ln for o in outs for ln in o[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["mode"] == "farm" and out["ranks"] == 2 and out["rank_sum"] == 3 and out["backend"] == "gloo"


def test_a_dying_rank_takes_the_launch_down_quickly():
    """a rank that exits before the rendezvous must not leave the others -- and the parent -- waiting for torch's time-out."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--multiprocess", "--steps", "3"], ERM_BENCH_DIE_RANK="1")
    assert r.returncode != 0 and time.time() - t0 < 120


@pytest.mark.gpu
def test_two_ranks_with_the_hip_engine_and_a_collective_on_one_gpu():
    """The N > 1 bench path with real engines: `bench.py --gpus 2` spawns two ranks, each runs its own chain through libertirt.so on cuda:0
    (ERM_BENCH_REHEARSE=1: a one-GPU box refuses two RCCL ranks on one device, so the collectives go over gloo), the ranks barrier around the
    timed region, MAX-reduce their times and all-reduce the posterior summaries."""
    e = dict(os.environ, ERM_BENCH_REHEARSE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "ERM_BENCH_LAUNCH_ONLY"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--multiprocess", "--steps", "12", "--warmup", "4", "--nsubj", "4000", "--nitem", "12",
                        "--clock-warmup-ms", "0"], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert out["n_gpus"] == 2 and out["config"]["chains"] == 2 and out["config"]["collective_backend"] == "gloo" and out["scaling"] == "weak"
    assert out["dtype"] == "f64" and out["fp32"]["dtype"] == "f32" and out["gather_ms"] > 0
    assert out["value"] == pytest.approx(4000 * 12 * 12 * 2 / (out["ms_per_step"] * 1e-3 * 12), rel=1e-6)      # whole-job aggregate over both ranks
    assert out["roofline"]["launches_timed"] >= 8 and out["roofline"]["algorithmic_bytes_per_cell_update"] == 25
    c4 = out["configs4"]                          # configs[4]'s per-GPU load (rehearsal size), data generated on the device
    assert c4["value"] > 0 and "nItem=100" in c4["workload"] and "device" in c4["data"]


@pytest.mark.gpu
def test_gpus_n_measures_the_c_abi_chain_farm():
    """`bench.py --gpus 2` = erm_farm_create(cfg, devices) / set_data / run / get_mean in ONE process (ERM_BENCH_REHEARSE=1 on a one-GPU box: both chains on
    device 0, the reduction forced through the library's RCCL communicator with one rank).  The line reports the library's communicator, the gather and
    per-chain device times, and BASELINE.json configs[4] as first-class fields."""
    e = dict(os.environ, ERM_BENCH_REHEARSE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "ERM_BENCH_LAUNCH_ONLY"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4", "--nsubj", "4000", "--nitem", "12",
                        "--clock-warmup-ms", "20"], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][0])
    c = out["config"]
    assert out["n_gpus"] == 2 and c["chains"] == 2 and c["devices"] == [0, 0] and "erm_farm_create" in c["path"] and out["scaling"] == "weak"
    assert c["rccl_ranks"] == 1 and out["gather"]["used_rccl"] and out["gather"]["n_devices"] == 1 and out["gather"]["post_count"] == 2 * 12
    assert out["gather_ms"] > 0 and out["allreduce_ms"] > 0 and len(out["per_chain_device_ms_per_step"]) == 2 and min(out["per_chain_device_ms_per_step"]) > 0
    assert out["dtype"] == "f64" and out["fp32"]["dtype"] == "f32" and out["ms_per_step_cold"] > 0
    assert out["value"] == pytest.approx(4000 * 12 * 12 * 2 / (out["ms_per_step"] * 1e-3 * 12), rel=1e-6)      # whole-job aggregate over both chains
    assert out["roofline"]["launches_timed"] >= 8 and out["roofline"]["algorithmic_bytes_per_cell_update"] == 25
    assert out["configs4_value"] == out["configs4"]["value"] > 0 and "nItem=100" in out["configs4"]["workload"] and "nChain=2" in out["configs4"]["workload"]
    assert len(out["configs4"]["per_chain_device_ms_per_step"]) == 2
    # the run validates itself (a multi-GPU run exits non-zero otherwise): the library's communicator spans the farm's devices (one here), the farm's
    # Post.mean is the count-weighted mean of the same chains run as separate engines, the per-chain timing check needs distinct devices and is left out
    sc = out["self_check"]
    assert sc["ok"] and sc["rccl_ranks"] == sc["rccl_ranks_expected"] == 1 and sc["farm_mean_vs_separate_engines_max_rel_err"] <= 1e-12 and sc["post_rows"] == sc["post_rows_expected"] == 2 * 16
    assert "not a BASELINE.json configuration" in c["workload"]
