"""The launch-geometry planner (extendedrtirtmodeling.jl_amd/csrc/erm_geometry.hpp) on the CPU: the header Engine::init calls is compiled by g++
with UndefinedBehaviorSanitizer and -ftrapv (division by zero, signed overflow and bad shifts abort) into tests/geometry_check.cpp's sweep over
N in 1 ... 2^32, J in 1 ... 896, F in 0 ... 14, every model and precision, the caller's overrides and odd compute-unit counts; every accepted plan
must satisfy the invariants the kernels rely on (LDS dynamic + static <= 160 KB, < 2^22 cells per workgroup, slices that hold the workgroup's
subjects, an 8-byte aligned accumulator region inside the allocation).  Named cases below are the geometry failures of earlier rounds."""
import os
import subprocess

import pytest

import parity_util as pu

SRC = os.path.join(pu.ROOT, "tests", "geometry_check.cpp")
INC = os.path.join(pu.ROOT, "extendedrtirtmodeling.jl_amd", "csrc")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("geom") / "geometry_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=all", "-ftrapv", "-I", INC, SRC, "-o", out], check=True)
    return out


def plan(exe, model, f64, N, J, Fk, bt=0, gb=0, W=0, cus=256, nofuse=0, nopersist=0):
    r = subprocess.run([exe, "case"] + [str(v) for v in (model, f64, N, J, Fk, bt, gb, W, cus, nofuse, nopersist)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    out = r.stdout.strip()
    if out.startswith("error="):
        return {"error": out[6:]}
    return dict(kv.split("=", 1) for kv in out.split())


def test_sweep_has_no_undefined_behaviour_and_every_plan_keeps_the_invariants(exe):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "invariant failures 0" in r.stdout and "(automatic geometry: 0)" in r.stdout, r.stdout


def test_static_lds_is_counted_between_158_and_160_kb(exe):
    """fp64 GibbsRtIrt around 175 000 x 50 -- the fused layout's dynamic LDS came within 2 KB of the limit while the kernel's static
    tables were not counted, so erm_create failed in hipFuncSetAttribute instead of taking the two-kernel schedule."""
    for N in range(150_000, 260_000, 1000):
        p = plan(exe, 1, 1, N, 50, 3)
        assert "error" not in p, (N, p)
        tot = int(p["lds_fused"] if p["fused"] == "1" else p["lds0"]) + int(p["lds_static"])
        assert tot <= 160 * 1024, (N, p)


def test_unsatisfiable_override_is_an_error_not_a_hang(exe):
    """block_threads = 1024 with 300 items (the per-wave accumulators alone exceed the LDS) made the grid-growth loop cycle between
    two counts for ever inside erm_create."""
    p = plan(exe, 1, 1, 100_000, 300, 3, bt=1024)
    assert "LDS footprint too large" in p["error"]
    assert "error" in plan(exe, 1, 1, 5000, 300, 3, bt=1024)


@pytest.mark.parametrize("case", [(1, 1, 64, 896, 1), (1, 1, 300, 5, 14), (3, 1, 300, 5, 14), (6, 1, 300, 5, 14), (0, 1, 300, 5, 14), (1, 1, 2, 3, 0), (2, 1, 3, 2, 0),
                                  (3, 1, 5, 1, 1), (1, 1, 130, 40, 3)])
def test_round2_geometry_failures_by_name(exe, case):
    """gpurun_out/r2_t2.txt (Floating point exception inside erm_create during test_f64_limits_of_the_engine: these are that test's sizes -- one
    of the divisors `nWaves - 1`, `grid_blocks`, `rows_per_block` was zero in the then planner), r2_t7.txt (64 x 896 x 1: LDS layout at the item limit),
    r2_full7.txt (130 x 40: an LDS request above the limit)."""
    p = plan(exe, *case)
    assert "error" not in p, p
    assert int(p["rows_per_block"]) >= 1 and int(p["grid_blocks"]) >= 1 and int(p["block_threads"]) >= 64
    for cu in (1, 8, 304):
        assert "error" not in plan(exe, *case, cus=cu)


def test_headline_geometries(exe):
    """What BASELINE.md's numbers were measured with: one workgroup of 1024 threads per CU (768 for fp64 LatentQr), fused; configs[4] in whole rounds."""
    p = plan(exe, 1, 1, 100_000, 50, 3)
    assert (p["block_threads"], p["grid_blocks"], p["fused"], p["rounds"]) == ("1024", "256", "1", "1")
    p = plan(exe, 3, 1, 100_000, 50, 3)
    assert (p["block_threads"], p["grid_blocks"], p["fused"]) == ("768", "256", "1")
    p = plan(exe, 1, 0, 100_000, 50, 3)
    assert (p["block_threads"], p["grid_blocks"], p["fused"]) == ("1024", "256", "1")
    p = plan(exe, 1, 1, 500_000, 100, 3)
    assert int(p["grid_blocks"]) > 256 and p["fused"] == "0"


def test_small_data_sets_get_the_persistent_schedule(exe):
    """Round 3: data sets of <= 2^17 cells and <= 128 items run all sweeps of an erm_run in ONE persistent launch: 32 workgroups of 512
    threads unless the caller gives a geometry, never more workgroups than CUs, only for the fused single-pass models; the per-sweep plan otherwise."""
    p = plan(exe, 0, 1, 1000, 15, 3)                         # configs[0]: GibbsMlIrt 1000 x 15
    assert (p["persist"], p["block_threads"], p["grid_blocks"], p["fused"], p["rounds"]) == ("1", "512", "32", "1", "1")
    assert plan(exe, 1, 0, 1000, 15, 3)["persist"] == "1" and plan(exe, 3, 1, 2000, 15, 3)["persist"] == "1"
    assert plan(exe, 1, 1, 4000, 30, 3)["grid_blocks"] == "64" and plan(exe, 1, 1, 4000, 30, 3)["persist"] == "1"      # 64 workgroups beyond ~1 500 subjects
    assert plan(exe, 1, 1, 8000, 16, 3)["persist"] == "0"                       # beyond 6 000 subjects the per-sweep schedule is faster
    assert plan(exe, 1, 1, 30, 5, 3)["persist"] == "1" and int(plan(exe, 1, 1, 30, 5, 3)["grid_blocks"]) <= 4
    assert plan(exe, 1, 1, 1000, 15, 3, nopersist=1)["persist"] == "0"          # ERM_FLAG_NO_PERSIST, sharded chains
    assert plan(exe, 1, 1, 1000, 15, 3, nofuse=1)["persist"] == "0"
    assert plan(exe, 2, 1, 1000, 15, 0)["persist"] == "0" and plan(exe, 5, 1, 1000, 15, 0)["persist"] == "0"      # two passes per sweep
    assert plan(exe, 1, 1, 100_000, 50, 3)["persist"] == "0" and plan(exe, 1, 1, 20_000, 12, 3)["persist"] == "0"   # beyond 2^17 cells
    assert plan(exe, 1, 1, 500, 200, 3)["persist"] == "0"                       # beyond 128 items
    assert plan(exe, 1, 1, 1000, 15, 3, bt=512, gb=32)["persist"] == "1"       # an explicit geometry that qualifies
    assert plan(exe, 1, 1, 1000, 15, 3, bt=1024, gb=32)["persist"] == "0" and plan(exe, 1, 1, 1000, 15, 3, bt=256, gb=125)["persist"] == "0"
    assert int(plan(exe, 1, 1, 1000, 15, 3, cus=8)["grid_blocks"]) <= 8 or plan(exe, 1, 1, 1000, 15, 3, cus=8)["persist"] == "0"
