"""Chain farm behind the C ABI (erm_farm_*, include/ertirt.h): nChain independent chains, one per device slot, sampled concurrently by
host threads of the library, Post.mean reduced on the device(s).  On the one-GPU test box every chain sits on device 0; the RCCL
all-reduce is exercised with a one-device communicator (erm_config.flags = ERM_FLAG_FARM_FORCE_RCCL)."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu


def _separate(model, Y, logT, X, inits, T, precision, qRt=0.85):
    """the same chains as independent engines: chain l = chain_id l, its own initial values"""
    L = pu.ge.load_package()._lib
    N, J = Y.shape
    engs = []
    for l, init in enumerate(inits):
        e = L.Engine(model=pu.MODELS[model], n_item=J, n_subj=N, n_feat=0 if X is None else X.shape[1], n_iter=T, n_chain=1, n_burnin=T // 2,
                     cov2one=int(model not in ("latentqr", "latent")), q_rt=qRt, seed=1234, chain_id=l, precision={"f32": 0, "f64": 1}[precision], trace_mode=1)
        e.set_data(Y, logT, X)
        e.set_state(**{("lambda_" if k == "lam" else k): v for k, v in init.items()})
        e.run(T)
        engs.append(e)
    return engs


@pytest.mark.parametrize("model", ["rtirt", "mlirt", "latentqr", "crossqr"])
@pytest.mark.parametrize("force_rccl", [False, True])
def test_farm_equals_separate_engines(model, force_rccl):
    L = pu.ge.load_package()._lib
    N, J, T, nch = 600, 8, 12, 3
    Y, logT, X, init, _ = pu.make_problem(model, N, J)
    g = np.random.default_rng(5)
    inits = [dict(init, theta=g.standard_normal(N)) for _ in range(nch)]
    farm = L.Farm([0] * nch, model=pu.MODELS[model], n_item=J, n_subj=N, n_feat=0 if X is None else X.shape[1], n_iter=T, n_chain=1, n_burnin=T // 2,
                  cov2one=int(model not in ("latentqr", "latent")), q_rt=0.85, seed=1234, precision=1, trace_mode=1, flags=L.FLAG_FARM_FORCE_RCCL if force_rccl else 0)
    farm.set_data(Y, logT, X)
    for l in range(nch):
        farm.set_state(l, **{("lambda_" if k == "lam" else k): v for k, v in inits[l].items()})
    farm.run(T)
    engs = _separate(model, Y, logT, X, inits, T, "f64")
    # traces: chain l in slab l, bit for bit the separate engine's
    for which in (L.TRACE_RA, L.TRACE_QR, L.TRACE_LOGLIKE) + (() if model == "mlirt" else (L.TRACE_RT,)):
        ft = farm.trace(which)
        assert ft.shape[2] == nch
        for l in range(nch):
            assert np.array_equal(ft[:, :, l], engs[l].trace(which)[:, :, 0]), (which, l)
    # Post.mean = joint mean over iterations and chains = count-weighted mean of the separate engines' means (to rounding)
    fm = farm.get_mean()
    assert farm.post_count == sum(e.post_count for e in engs) and farm.used_rccl == force_rccl
    tm = farm.timing()          # erm_farm_get_timing: what bench.py --gpus N reports
    assert tm["rccl_ranks"] == (1 if force_rccl else 0) and tm["n_devices"] == 1 and tm["run_wall_ms"] > 0 and tm["gather_ms"] > 0 and np.all(tm["run_ms"] > 0)
    assert (tm["allreduce_ms"] > 0) == force_rccl
    means = [e.get_mean() for e in engs]
    for k, v in fm.items():
        if v is None:
            continue
        want = sum(m[k] * e.post_count for m, e in zip(means, engs)) / farm.post_count
        assert np.max(np.abs(v - want) / np.maximum(np.abs(want), 1e-12)) < 1e-13, k
    # chains differ (independent streams) and the farm can continue
    assert not np.array_equal(farm.trace(L.TRACE_RA)[:, :, 0], farm.trace(L.TRACE_RA)[:, :, 1])
    st = farm.get_state(1)
    assert np.array_equal(st["theta"], engs[1].get_state()["theta"])


def test_farm_through_the_sample_mirror_and_errors():
    pkg = pu.ge.load_package()
    L = pkg._lib
    Cond = pkg.setCond(nSubj=500, nItem=7, nIter=20, nChain=2)
    g = np.random.default_rng(3)
    tp = pkg.setTrueParaRtIrt(Cond, seed=g)
    Data = pkg.setDataRtIrt(Cond, tp, seed=g)
    M = pkg.GibbsRtIrt(Cond, Data=Data, truePara=tp)
    pkg.sample_b(M, devices=[0])
    assert M.Post.ra.shape == (20, 500 + 14, 2) and M.Post.logLike.shape == (20, 1, 2)
    assert np.all(np.isfinite(M.Post.ra)) and M.Post.mean.theta.shape == (500,)
    # joint mean over the post-burn-in iterations of both chains (src/GibbsRtIrt.pl.jl:327-343)
    want = M.Post.ra[Cond.nBurnin:, 500:507, :].mean(axis=(0, 2))
    assert np.max(np.abs(M.Post.mean.a - want)) < 1e-12
    with pytest.raises(L.ErmError, match="no such device"):
        L.Farm([0, 99], model=1, n_item=7, n_subj=500, n_feat=3, n_iter=4, n_chain=1, n_burnin=2, cov2one=1, q_rt=0.5, seed=1, precision=1, trace_mode=0)
    f = L.Farm([0, 0], model=1, n_item=7, n_subj=500, n_feat=3, n_iter=4, n_chain=1, n_burnin=2, cov2one=1, q_rt=0.5, seed=1, precision=1, trace_mode=0)
    with pytest.raises(L.ErmError, match="chain 0"):
        f.run(1)                                   # no data yet: the chain's own error, named
    with pytest.raises(L.ErmError, match="no post-burn-in"):
        f.get_mean()


def test_farm_runs_are_reproducible():
    """Twenty farms in a row (three host threads inside erm_set_data / erm_run each time) give chain 0 the same trace: a fill or an upload on the NULL
    stream that is not complete when an engine's own non-blocking stream starts working shows up here (it did, about once in 500 farm runs)."""
    L = pu.ge.load_package()._lib
    N, J, T, nch = 600, 8, 6, 3
    Y, logT, X, init, _ = pu.make_problem("rtirt", N, J)
    st = {("lambda_" if k == "lam" else k): v for k, v in init.items()}
    ref = None
    for _ in range(20):
        farm = L.Farm([0] * nch, model=pu.MODELS["rtirt"], n_item=J, n_subj=N, n_feat=X.shape[1], n_iter=T, n_chain=1, n_burnin=T // 2, cov2one=1, q_rt=0.85,
                      seed=1234, precision=1, trace_mode=1)
        farm.set_data(Y, logT, X)
        for l in range(nch):
            farm.set_state(l, **st)
        farm.run(T)
        tr = farm.trace(L.TRACE_RA).copy()
        if ref is None:
            ref = tr
        assert np.array_equal(tr, ref)
