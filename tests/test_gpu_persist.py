"""Small data sets: ONE persistent launch per erm_run (pass_kernel<..., PERSIST>: a grid barrier between the sweeps instead of a kernel boundary).
The chain is the same chain: at one geometry the persistent schedule, the per-sweep launches (ERM_FLAG_NO_PERSIST) and the per-sweep launches without
graphs give bit-identical traces; against the oracle the usual fp64 bound holds."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu

L = pu.ge.load_package()._lib
GEOM = dict(block_threads=512, grid_blocks=32)


def _same(a, b, keys=("ra", "item", "ll", "qr")):
    return all(np.array_equal(a[k], b[k]) for k in keys if k in a)


@pytest.mark.parametrize("model", ["mlirt", "rtirt", "null", "latent", "latentqr"])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_persistent_launch_is_the_per_sweep_chain(model, precision):
    Y, logT, X, init, _ = pu.make_problem(model, 1000, 15)
    T = 45
    per = pu.run_device(model, Y, logT, X, init, T, precision=precision, **GEOM)
    assert per["engine"].timing()["persistent"] == 1
    for flags in (L.FLAG_NO_PERSIST, L.FLAG_NO_PERSIST | L.FLAG_NO_GRAPH):
        ref = pu.run_device(model, Y, logT, X, init, T, precision=precision, flags=flags, **GEOM)
        assert ref["engine"].timing()["persistent"] == 0
        assert _same(per, ref), (model, precision, flags)
        if model != "mlirt":
            assert np.array_equal(per["rt"], ref["rt"])
        for k, v in per["state"].items():
            assert np.array_equal(np.asarray(v), np.asarray(ref["state"][k])), k


def test_default_geometry_of_a_small_data_set_is_persistent_and_matches_the_oracle():
    d = pu.run_pair("rtirt", 1000, 15, 12)
    assert d["dev"]["engine"].timing()["persistent"] == 1
    assert np.max(np.abs(d["dev_ra"] - d["orc"]["ra"]) / (1 + np.abs(d["orc"]["ra"]))) < 1e-8


def test_split_runs_continue_the_chain():
    """erm_run(7) + erm_run(1) + erm_run(12) = erm_run(20): the launch starts from whichever half of the double buffers is current."""
    Y, logT, X, init, _ = pu.make_problem("rtirt", 600, 10)
    one = pu.run_device("rtirt", Y, logT, X, init, 20, precision="f64")
    eng = L.Engine(model=pu.MODELS["rtirt"], n_item=10, n_subj=600, n_feat=3, n_iter=20, n_chain=1, n_burnin=10, cov2one=1, q_rt=0.85, seed=1234, precision=1, trace_mode=1)
    eng.set_data(Y, logT, X)
    eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in init.items()})
    for n in (7, 1, 12):
        eng.run(n)
    assert eng.timing()["persistent"] == 1
    assert np.array_equal(eng.item_trace(), one["item"]) and np.array_equal(eng.trace(L.TRACE_RA), one["ra"])


def test_larger_data_sets_and_sharded_chains_keep_the_per_sweep_schedule():
    Y, logT, X, init, _ = pu.make_problem("rtirt", 20000, 12)      # 240k cells > the persistent limit
    out = pu.run_device("rtirt", Y, logT, X, init, 4, precision="f64")
    assert out["engine"].timing()["persistent"] == 0
    for model in ("cross", "crossqr"):                              # two row passes per sweep with item draws between them: per-sweep launches
        Y, logT, X, init, _ = pu.make_problem(model, 500, 8)
        assert pu.run_device(model, Y, logT, X, init, 4, precision="f64")["engine"].timing()["persistent"] == 0


def test_two_engines_on_one_device_take_turns():
    """Two persistent launches at once on one device could each wait for the other's compute units: the library serialises them per device."""
    import threading
    Y, logT, X, init, _ = pu.make_problem("mlirt", 1000, 15)
    ref = pu.run_device("mlirt", Y, logT, X, init, 400, precision="f64", trace_full=False)
    outs = [None, None]

    def work(k):
        outs[k] = pu.run_device("mlirt", Y, logT, X, init, 400, precision="f64", trace_full=False)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert all(not t.is_alive() for t in th)
    for o in outs:
        assert np.array_equal(o["item"], ref["item"])


def test_persistent_runs_are_reproducible_over_long_chains():
    """The packet exchange has no barrier to hide a race behind: 300-sweep chains in one launch, repeated, bit for bit against the per-sweep chain
    (tools/persist_stress.py is the long form: 1 200 runs over six shapes and both precisions, no mismatch)."""
    for model, N, J in (("rtirt", 1000, 15), ("latent", 250, 100), ("mlirt", 37, 5)):
        Y, logT, X, init, _ = pu.make_problem(model, N, J)
        per = pu.run_device(model, Y, logT, X, init, 300, precision="f64", trace_full=False)
        tm = per["engine"].timing()
        assert tm["persistent"] == 1
        ref = pu.run_device(model, Y, logT, X, init, 300, precision="f64", trace_full=False, flags=L.FLAG_NO_PERSIST, block_threads=tm["block_threads"], grid_blocks=tm["grid_blocks"])
        for _ in range(6):
            got = pu.run_device(model, Y, logT, X, init, 300, precision="f64", trace_full=False)
            assert np.array_equal(got["item"], ref["item"]) and np.array_equal(got["ll"], ref["ll"]), (model, N, J)


@pytest.mark.parametrize("model,precision", [("rtirt", "f64"), ("mlirt", "f32"), ("latentqr", "f64")])
def test_a_persistent_launch_that_times_out_is_replayed_per_sweep(model, precision):
    """ERM_FLAG_TEST_PERSIST_TIMEOUT makes workgroup 1 of the engine's second persistent erm_run lose its statistics row and shortens the wait to 2 ms:
    the launch times out, every workgroup leaves it, and erm_run restores the state it saved, drops the persistent schedule and replays the call
    one launch per sweep.  The call SUCCEEDS, erm_timing says what happened, and the chain is the chain of an undisturbed engine bit for bit --
    also across a run that continues it, and with post-burn-in sums (Post.mean) that were saved and restored."""
    Y, logT, X, init, _ = pu.make_problem(model, 1000, 15)
    good = pu.run_device(model, Y, logT, X, init, 40, precision=precision, n_burnin=5)
    assert good["engine"].timing()["persistent"] == 1 and good["engine"].timing()["persist_fallbacks"] == 0
    eng = L.Engine(model=pu.MODELS[model], n_item=15, n_subj=1000, n_feat=3, n_iter=40, n_chain=1, n_burnin=5, cov2one=int(model != "latentqr"), q_rt=0.85, seed=1234,
                   precision={"f32": 0, "f64": 1}[precision], trace_mode=1, flags=L.FLAG_TEST_PERSIST_TIMEOUT)
    assert eng.timing()["persistent"] == 1
    eng.set_data(Y, logT, X)
    eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in init.items()})
    eng.run(8)                                    # a persistent run that succeeds: post-burn-in rows and sums exist before the failing call
    tm = eng.timing()
    assert tm["persistent"] == 1 and tm["persist_fallbacks"] == 0
    eng.run(20)                                   # the engine's second persistent run loses a packet: time-out, restore, per-sweep replay
    tm = eng.timing()
    assert tm["persistent"] == 0 and tm["persist_fallbacks"] == 1
    eng.run(12)
    assert eng.timing()["persist_fallbacks"] == 1
    assert np.array_equal(eng.item_trace(), good["item"]) and np.array_equal(eng.trace(L.TRACE_RA), good["ra"]) and np.array_equal(eng.trace(L.TRACE_LOGLIKE), good["ll"])
    gm, em = good["engine"].get_mean(), eng.get_mean()
    for k, v in gm.items():
        assert v is None or np.array_equal(v, em[k]), k


def test_schedule_flags_keep_the_default_geometry():
    """ERM_FLAG_NO_PERSIST changes the schedule, not the launch geometry: with nothing pinned the per-sweep engine runs the persistent plan's workgroups
    (erm_geometry.hpp), sums its statistics in the same association and draws the same chain bit for bit."""
    for model, N, J in (("rtirt", 1000, 15), ("mlirt", 300, 20)):
        Y, logT, X, init, _ = pu.make_problem(model, N, J)
        per = pu.run_device(model, Y, logT, X, init, 30, precision="f64")
        ref = pu.run_device(model, Y, logT, X, init, 30, precision="f64", flags=L.FLAG_NO_PERSIST)
        tp, tr = per["engine"].timing(), ref["engine"].timing()
        assert (tp["persistent"], tr["persistent"]) == (1, 0)
        assert (tp["block_threads"], tp["grid_blocks"]) == (tr["block_threads"], tr["grid_blocks"])
        assert _same(per, ref)
