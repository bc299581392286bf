"""runSimulation (/root/reference/src/SimTools.jl:457-495) end to end on the device: ONE engine per condition, every replication's data
set generated on the device, DIC and convergence counts computed there -- only summaries cross the boundary."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu


def _forbid(monkeypatch, L, names):
    def boom(name):
        def f(*a, **k):
            raise AssertionError(f"runSimulation(on_device) must not call Engine.{name}")
        return f
    for n in names:
        monkeypatch.setattr(L.Engine, n, boom(n))


def test_run_simulation_stays_on_the_device(monkeypatch):
    """nRep = 5 at 1 000 x 15 (the reference's own size): the replications never upload a data set, pull a trace, the item trace, the data or the N-wide
    diagnostics; what comes back per replication is Post.mean of the requested fields, the DIC, checkConvergence's counters and the wall time."""
    pkg = pu.ge.load_package()
    L = pkg._lib
    Cond = pkg.setCond(nSubj=1000, nItem=15, nFeat=3, nIter=300, nChain=2, nRep=5)
    tp = pkg.setTrueParaRtIrt(Cond, seed=3)
    _forbid(monkeypatch, L, ("set_data", "get_data", "trace", "item_trace", "diagnostics"))
    Run = pkg.runSimulation(Cond, tp, Para=("a", "b", "λ", "σ²t", "β", "Σp"), seed=100)
    assert set(Run) == {"True", 1, 2, 3, 4, 5}
    secs = []
    for r in range(1, 6):
        P = Run[r]
        assert P["a"].shape == (15,) and P["β"].shape == (8,) and P["Σp"].shape == (4,) and np.isfinite(P["Dic"][0])
        assert set(P["Diag"]) == {"ess", "rhat", "essN", "rhatN"} and 0 <= P["Diag"]["rhat"] <= 100
        secs.append(P["Seconds"])
    # five replications of 600 sweeps on 15 000 cells: recovery in the reference's own validation style (README.md:61-77; the discriminations mix slowly,
    # their bound is loose)
    m = pkg.getMetrics(Run, par="a")
    print("a:", m, "b:", pkg.getMetrics(Run, par="b"), "lambda:", pkg.getMetrics(Run, par="λ"))
    assert m["Rmse"] < 0.25 and m["Corr"] > 0.6
    # (lambda_j - zeta_i is identified up to a common shift only by zeta's prior: the intensities come back with a slowly mixing common offset)
    assert pkg.getMetrics(Run, par="b")["Rmse"] < 0.15 and pkg.getMetrics(Run, par="b")["Corr"] > 0.9
    assert pkg.getMetrics(Run, par="λ")["Corr"] > 0.97 and abs(pkg.getMetrics(Run, par="λ")["Bias"]) < 0.4
    assert len({tuple(np.round(Run[r]["a"], 12)) for r in range(1, 6)}) == 5          # five different data sets, five different chains
    print("per-replication wall time (s):", [round(s, 3) for s in secs])
    assert max(secs[1:]) < 5.0


def test_device_replication_equals_the_same_replication_through_the_host():
    """One replication of the device path against the same steps taken apart: the data set of that replication pulled from the device, uploaded into a FRESH
    sampler with the replication's seed, sample!, host-side Post.mean / getDicHost / checkConvergence(detail).  Same chain (1e-9: the upload re-derives the data
    constants from the pulled values), same DIC, same counters."""
    pkg = pu.ge.load_package()
    L = pkg._lib
    Cond = pkg.setCond(nSubj=600, nItem=10, nFeat=2, nIter=60, nChain=1, nRep=2)
    tp = pkg.setTrueParaRtIrt(Cond, seed=5)
    seed = 40
    Run = pkg.runSimulation(Cond, tp, Para=("a", "σ²t", "β"), seed=seed)
    run = 2
    M0 = pkg.GibbsRtIrt(Cond, truePara=tp, seed=seed + run)
    import copy
    pkg.simulateData(M0, copy.copy(tp), seed=int(np.random.SeedSequence([seed, run]).generate_state(1, dtype=np.uint64)[0]), pull=True)
    M = pkg.GibbsRtIrt(Cond, truePara=tp, Data=M0.Data, seed=seed + run)
    M0.close()
    pkg.sample_b(M)
    assert np.max(np.abs(M.Post.mean.a - Run[run]["a"])) < 1e-9 and np.max(np.abs(M.Post.mean.sig2t - Run[run]["σ²t"])) < 1e-9
    assert np.max(np.abs(M.Post.mean.beta.reshape(-1, order="F") - Run[run]["β"])) < 1e-9
    assert abs(pkg.getDicHost(M).DIC - Run[run]["Dic"][0]) < 1e-7 * abs(Run[run]["Dic"][0])
    host = pkg.checkConvergence(M)
    assert (host["essN"], host["rhatN"]) == (Run[run]["Diag"]["essN"], Run[run]["Diag"]["rhatN"])
    M.close()


@pytest.mark.parametrize("gibbs,data,truth,par", [("GibbsMlIrt", "setDataMlIrt", "setTrueParaMlIrt", ("a", "b")),
                                                    ("GibbsRtIrtLatentQr", "setDataRtIrtLatent", "setTrueParaRtIrtLatent", ("a", "λ", "β")),
                                                    ("GibbsRtIrtCrossQr", "setDataRtIrtCross", "setTrueParaRtIrtCross", ("a", "ρ")),
                                                    ("GibbsRtIrtNull", "setDataRtIrtNull", "setTrueParaRtIrt", ("b", "Σp"))])
def test_run_simulation_on_device_for_the_other_generators(gibbs, data, truth, par):
    pkg = pu.ge.load_package()
    Cond = pkg.setCond(nSubj=400, nItem=8, nFeat=2, nIter=40, nChain=1, nRep=2, qRt=0.85)
    tp = getattr(pkg, truth)(Cond, seed=9)
    if gibbs == "GibbsRtIrtLatentQr":
        tp.sig2t = np.ones(Cond.nItem)
    Run = pkg.runSimulation(Cond, tp, Para=par, funcData=getattr(pkg, data), funcGibbs=getattr(pkg, gibbs), seed=7, typeName="norm")
    for r in (1, 2):
        assert all(np.all(np.isfinite(Run[r][p])) for p in par) and np.isfinite(Run[r]["Dic"][0])
    # a user-supplied generator cannot be restated on the device: the host path is taken (and refused when the device path is demanded)
    custom = lambda C, t, **kw: getattr(pkg, data)(C, t, **kw)
    R2 = pkg.runSimulation(Cond, tp, Para=par, funcData=custom, funcGibbs=getattr(pkg, gibbs), seed=7)
    assert set(R2) == {"True", 1, 2}
    with pytest.raises(ValueError, match="on_device"):
        pkg.runSimulation(Cond, tp, Para=par, funcData=custom, funcGibbs=getattr(pkg, gibbs), seed=7, on_device=True)
