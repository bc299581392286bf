"""Error behaviour of the C-ABI boundary on a live GPU: every misuse returns a negative code + message (never a crash, a hang or a
silent fallback), mirroring the reference's own checks where it has them (@assert 0 < qRt < 1 and nu > 0, src/Draw.pl.jl:476-477;
itemtype validation is host-side, tests/test_host_api.py)."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu
L = pu.ge.load_package()._lib


def _engine(model="rtirt", N=200, J=6, F=2, **kw):
    args = dict(model=pu.MODELS.get(model, model), n_item=J, n_subj=N, n_feat=F, n_iter=4, n_chain=1, n_burnin=2, cov2one=1, q_rt=0.85, seed=1,
                precision=1, trace_mode=1)
    args.update(kw)
    return L.Engine(**args)


def _data(N=200, J=6, F=2, seed=0):
    g = np.random.default_rng(seed)
    return (g.random((N, J)) < 0.5).astype(np.uint8), g.normal(3, 0.5, (N, J)), g.standard_normal((N, F))


@pytest.mark.parametrize("kw,msg", [
    (dict(n_item=0), "positive"), (dict(n_subj=0), "positive"), (dict(n_item=897), "n_item too large"), (dict(n_feat=15), "n_feat too large"),
    (dict(model=9), "unknown model"), (dict(precision=7), "precision"), (dict(block_threads=100), "block_threads"),
    (dict(lanes_per_row=3), "lanes_per_row"), (dict(sigp_mode=2), "sigp_mode"), (dict(n_chain=0), "n_chain")])
def test_create_rejects_bad_configurations(kw, msg):
    with pytest.raises(L.ErmError, match=msg):
        _engine(**kw)


@pytest.mark.parametrize("model", ["crossqr", "latentqr"])
@pytest.mark.parametrize("q", [0.0, 1.0, -0.3])
def test_quantile_level_must_be_inside_the_unit_interval(model, q):      # @assert at src/Draw.pl.jl:476
    with pytest.raises(L.ErmError, match="qRt"):
        _engine(model=model, q_rt=q)


def test_set_data_validates_its_inputs():
    Y, logT, X = _data()
    e = _engine()
    with pytest.raises((ValueError, L.ErmError), match="0/1"):      # the ctypes wrapper checks first; the library checks again
        e.set_data(Y * 2, logT, X)
    bad = logT.copy(); bad[3, 2] = np.inf
    with pytest.raises(L.ErmError, match="finite"):
        e.set_data(Y, bad, X)
    Xc = X.copy(); Xc[:, 1] = 2 * Xc[:, 0]
    with pytest.raises(L.ErmError, match="singular"):
        e.set_data(Y, logT, Xc)
    with pytest.raises(L.ErmError, match="logT is required"):
        e.set_data(Y, None, X)
    with pytest.raises(L.ErmError, match="erm_set_data has not been called"):
        e.run(1)
    e.set_data(Y, logT, X)
    e.run(4)


def test_state_and_trace_misuse():
    Y, logT, X = _data()
    e = _engine()
    e.set_data(Y, logT, X)
    with pytest.raises(L.ErmError, match="sig2t must be positive"):
        e.set_state(sig2t=np.zeros(6))
    e.run(2)
    with pytest.raises(L.ErmError, match="trace incomplete"):
        e.trace(L.TRACE_RA)
    with pytest.raises(L.ErmError, match="no post-burn-in"):
        e.get_mean()
    e.run(2)
    assert e.trace(L.TRACE_RA).shape == (4, 200 + 12, 1) and e.post_count == 2
    with pytest.raises(L.ErmError, match="trace capacity exceeded"):
        e.run(1)
    e.reset_trace()
    e.run(4)                                                    # the chain continues after a reset
    s = _engine(trace_mode=0)
    s.set_data(Y, logT, X)
    s.run(4)
    with pytest.raises(L.ErmError, match="ERM_TRACE_FULL"):
        s.trace(L.TRACE_RA)
    assert s.trace(L.TRACE_LOGLIKE).shape == (4, 1, 1) and s.item_trace().shape[0] == 4
    cq = _engine(model="crossqr", F=0)
    cq.set_data(Y, logT, None)
    with pytest.raises(L.ErmError, match="nu must be positive"):      # @assert at src/Draw.pl.jl:477
        cq.set_state(nu=np.zeros((200, 6)))


def test_a_poisoned_state_is_reported_not_propagated():
    """A non-finite parameter sets the sticky device flag: erm_run names the entry instead of returning NaN traces."""
    Y, logT, X = _data()
    e = _engine()
    e.set_data(Y, logT, X)
    e.set_state(theta=np.full(200, np.nan))
    with pytest.raises(L.ErmError, match="non-finite parameter"):
        e.run(4)


def test_stage_timing_knobs_do_not_exist_in_the_shipped_library(monkeypatch):
    """ERM_PASS_STOP / ERM_TINY_STOP / ERM_SKEW are stage-timing knobs of a -DERM_DIAG_BUILD library (tools/tiny_stages.sh): early returns
    that leave garbage results.  The shipped library compiles none of that code, so a variable left exported cannot touch a fit."""
    Y, logT, X, init, _ = pu.make_problem("rtirt", 700, 9)
    ref = pu.run_device("rtirt", Y, logT, X, init, 6, precision="f64")
    for name, val in (("ERM_PASS_STOP", "3"), ("ERM_PASS_STOP", "9"), ("ERM_TINY_STOP", "1"), ("ERM_SKEW", "-7"), ("ERM_SKEW", "100000"), ("ERM_STOP_SWEEP", "2"),
                      ("ERM_NO_FUSE", "1"), ("ERM_NO_GRAPH", "1"), ("ERM_FARM_FORCE_RCCL", "1"), ("ERM_NU_TRACE_MAX_GB", "0.000001"), ("ERM_TIMELINE", "1")):
        monkeypatch.setenv(name, val)
        for prec in ("f64", "f32"):
            got = pu.run_device("rtirt", Y, logT, X, init, 6, precision=prec)
            if prec == "f64":
                assert np.array_equal(got["ra"], ref["ra"]) and np.array_equal(got["item"], ref["item"]) and np.array_equal(got["ll"], ref["ll"]), (name, val)
            else:
                assert np.all(np.isfinite(got["ra"])) and np.max(np.abs(got["item"] - ref["item"])) < 5e-2, (name, val)
        monkeypatch.delenv(name)


@pytest.mark.parametrize("model", ["rtirt", "mlirt", "latentqr"])
def test_schedule_flags_change_the_schedule_not_the_chain(model):
    """erm_config.flags (round 3: schedule switches are config fields, not environment variables): ERM_FLAG_NO_FUSE = stand-alone tiny kernel + row pass
    instead of the fused sweep kernel, ERM_FLAG_NO_GRAPH = every sweep enqueued instead of the replayed 32-sweep hipGraph, ERM_FLAG_NO_PERSIST = per-sweep launches instead of one
    persistent launch per erm_run (small data sets).  Same chain: item-level draws
    agree to rounding of the statistics' summation (the fused head and the tiny kernel reduce the same rows in the same order: bit-identical)."""
    Y, logT, X, init, _ = pu.make_problem(model, 900, 9)
    T = 40                                    # long enough for the 32-sweep graph
    geom = dict(block_threads=512, grid_blocks=32)       # one geometry for every schedule (a data set this small defaults to the persistent launch and ITS geometry)
    ref = pu.run_device(model, Y, logT, X, init, T, precision="f64", **geom)
    for flags in (L.FLAG_NO_GRAPH, L.FLAG_NO_PERSIST, L.FLAG_NO_PERSIST | L.FLAG_NO_GRAPH, L.FLAG_NO_FUSE, L.FLAG_NO_FUSE | L.FLAG_NO_GRAPH):
        got = pu.run_device(model, Y, logT, X, init, T, precision="f64", flags=flags, **geom)
        assert np.array_equal(got["ra"], ref["ra"]) and np.array_equal(got["item"], ref["item"]) and np.array_equal(got["ll"], ref["ll"]), flags


def test_nu_trace_budget_is_a_config_field():
    """GibbsRtIrtCrossQr's Post.qr carries vec(nu) per sweep; erm_config.nu_trace_max_gb bounds the device memory it may take (default 16 GiB)."""
    Y, logT, X, init, _ = pu.make_problem("crossqr", 200, 6)
    full = pu.run_device("crossqr", Y, logT, X, init, 4, precision="f64")
    assert full["qr"].shape[1] == 6 + 4 + 200 * 6
    eng = L.Engine(model=pu.MODELS["crossqr"], n_item=6, n_subj=200, n_feat=0, n_iter=4, n_chain=1, n_burnin=2, cov2one=1, q_rt=0.85, seed=1234, precision=1, trace_mode=1,
                   nu_trace_max_gb=1e-7)
    eng.set_data(Y, logT, None)
    eng.set_state(**init)
    eng.run(4)
    with pytest.raises(L.ErmError, match="nu trace"):
        eng.trace(L.TRACE_QR)
    assert np.array_equal(eng.item_trace(), full["item"])


def test_a_failed_run_poisons_the_engine_until_a_state_is_installed():
    """erm_run failing part-way (here: the shard exchange callback raising on its second call) must not leave the double buffers in
    an unknown parity silently: the engine refuses erm_run / erm_get_state until erm_set_state."""
    Y, logT, X, init, _ = pu.make_problem("rtirt", 300, 6)
    eng = L.Engine(model=pu.MODELS["rtirt"], n_item=6, n_subj=300, n_feat=3, n_iter=8, n_chain=1, n_burnin=4, cov2one=1, q_rt=0.85, seed=1234, precision=1, trace_mode=1)
    lib = L.load()
    calls = {"n": 0, "fail_at": None}

    def exchange(send, recv, nbytes):          # a one-shard "all-gather": copy send -> recv
        calls["n"] += 1
        if calls["fail_at"] is not None and calls["n"] >= calls["fail_at"]:
            raise RuntimeError("link down")
        assert lib.erm_copy(recv, send, nbytes) == 0
    eng.set_shard(0, 1, 300, 0, exchange)
    eng.set_data(Y, logT, X)
    st = {("lambda_" if k == "lam" else k): v for k, v in init.items()}
    eng.set_state(**st)
    eng.run(2)
    calls["fail_at"] = calls["n"] + 2
    with pytest.raises(L.ErmError, match="exchange"):
        eng.run(3)
    with pytest.raises(L.ErmError, match="failed part-way"):
        eng.run(1)
    with pytest.raises(L.ErmError, match="failed part-way"):
        eng.get_state()
    calls["fail_at"] = None
    eng.set_state(**st)
    eng.reset_trace()
    eng.run(2)
    assert np.all(np.isfinite(eng.item_trace()))
