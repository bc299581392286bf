"""HIP path against the committed golden fixtures (tests/golden/*.npz: inputs + oracle traces), through the Julia-surface mirror
(Gibbs* constructors, sample!, Post) where the model allows it."""
import numpy as np
import pytest

import parity_util as pu
from test_oracle_sweeps import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model", ["mlirt", "rtirt", "latentqr", "crossqr", "null", "cross", "latent"])
def test_engine_reproduces_golden_traces(model):
    z, Y, logT, X, init = load_golden(model)
    T = int(z["T"]) if model != "crossqr" else 3
    dev = pu.run_device(model, Y, logT, X, init, T, precision="f64", qRt=float(z["qRt"]), seed=int(z["seed"]))
    assert pu.rel_err(dev["ra"][:, :, 0], z["ra"][:T]).max() < 1e-8
    if model != "mlirt":
        assert pu.rel_err(dev["rt"][:, :, 0], z["rt"][:T]).max() < 1e-8
    wq = z["qr"].shape[1]              # CrossQr's golden holds [rho; vec(Sigp)]; the device row continues with vec(nu)
    assert pu.rel_err(dev["qr"][:, :wq, 0], z["qr"][:T]).max() < 1e-8
    assert pu.rel_err(dev["ll"][:, 0, 0], z["ll"][:T]).max() < 1e-8


def test_sample_bang_fills_post_like_the_reference():
    """GibbsRtIrt(Cond; Data) |> sample! : Post.ra/rt/qr/logLike shapes and layout (src/GibbsRtIrt.pl.jl:63-70), Post.mean flat
    vectors over m > nBurnin and all chains (:327-343), Para left at the final state, second call continues the chain."""
    pkg = pu.ge.load_package()
    z, Y, logT, X, init = load_golden("rtirt")
    N, J = Y.shape
    Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=3, nIter=6, nChain=2, qRt=0.5)
    M = pkg.GibbsRtIrt(Cond, Data=pkg.InputData(Y=Y, T=np.exp(logT), X=X), precision="f64")
    assert pkg.sample_b(M) is M
    P = M.Post
    assert P.ra.shape == (6, N + 2 * J, 2) and P.rt.shape == (6, N + 2 * J, 2) and P.qr.shape == (6, 12, 2) and P.logLike.shape == (6, 1, 2)
    assert np.allclose(P.mean.theta, P.ra[3:, :N, :].mean(axis=(0, 2))) and np.allclose(P.mean.sig2t, P.rt[3:, N + J:, :].mean(axis=(0, 2)))
    assert np.allclose(P.mean.beta, P.qr[3:, :8, :].mean(axis=(0, 2))) and P.mean.Sigp.shape == (4,)
    assert np.array_equal(M.Para.theta, P.ra[5, :N, 1]) and np.array_equal(M.Para.b, P.ra[5, N + J:, 1])
    assert np.all(P.qr[:, 0, :] == 0) and np.all(P.qr[:, 4, :] == 0)          # intercept=false
    assert np.all(P.qr[:, 8, :] == 1) and np.all(P.qr[:, 11, :] == 1)         # cov2one=true
    first = P.ra.copy()
    pkg.sample_b(M)
    assert not np.array_equal(first, M.Post.ra)                                # continues from Para with fresh variates
    d = pkg.getDic(M)
    assert np.isfinite(d.DIC) and np.isfinite(d.pD)
    assert set(pkg.coef(M)) >= {"a", "b", "β", "Σp"} and len(pkg.precis(M)["names"]) == 4 * J + 12
    M1 = pkg.GibbsRtIrt(Cond, Data=M.Data, precision="f64")
    pkg.sample_b(M1, itemtype="1pl")
    assert np.all(M1.Post.ra[:, N:N + J, :] == 1)


@pytest.mark.parametrize("name,model", [("GibbsRtIrtNull", "null"), ("GibbsRtIrtCross", "cross"), ("GibbsRtIrtLatent", "latent")])
def test_variant_samplers_fill_post_like_the_reference(name, model):
    """GibbsRtIrtNull / Cross / Latent through the Julia-surface mirror: Post widths of OutputPost / OutputPostCross /
    OutputPostRtIrtLatent (src/GibbsRtIrt.pl.jl:67, src/GibbsRtIrtCross.pl.jl:46, src/GibbsRtIrtLatent.pl.jl:43), Post.mean fields
    of each sample! (:407-423, :216-232, :214-230), DIC."""
    pkg = pu.ge.load_package()
    z, Y, logT, X, init = load_golden(model)
    N, J = Y.shape
    Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=3, nIter=6, nChain=2)
    D = pkg.InputData(Y=Y, T=np.exp(logT), X=X if X is not None else np.zeros((N, 3)))
    M = getattr(pkg, name)(Cond, Data=D, precision="f64")
    pkg.sample_b(M)
    P = M.Post
    wq = {"null": 12, "cross": J + 4, "latent": 9}[model]
    assert P.ra.shape == (6, N + 2 * J, 2) and P.rt.shape == (6, N + 2 * J, 2) and P.qr.shape == (6, wq, 2) and P.logLike.shape == (6, 1, 2)
    assert np.allclose(P.mean.theta, P.ra[3:, :N, :].mean(axis=(0, 2))) and np.allclose(P.mean.lam, P.rt[3:, N:N + J, :].mean(axis=(0, 2)))
    if model == "null":
        assert np.all(P.qr[:, :8, :] == 0) and np.all(P.mean.beta == 0) and np.allclose(P.mean.Sigp, P.qr[3:, 8:, :].mean(axis=(0, 2)))
    if model == "cross":
        assert np.allclose(P.mean.rho, P.qr[3:, :J, :].mean(axis=(0, 2))) and np.all(P.qr[:, J:, :] == np.array([1, 0, 0, 1])[None, :, None])
    if model == "latent":
        assert np.allclose(P.mean.beta, P.qr[3:, :5, :].mean(axis=(0, 2))) and np.all(P.qr[:, 0, :] == 0) and np.all(P.qr[:, 5, :] == 1)
        assert not np.all(P.qr[:, 8, :] == 1)                                  # cov2one defaults to false for Latent
    d = pkg.getDic(M)
    assert np.isfinite(d.DIC) and np.isfinite(d.pD)


def test_crossqr_post_qr_carries_rho_sigp_and_vec_nu():
    """Post.qr of GibbsRtIrtCrossQr is [rho; vec(Sigp); vec(nu)] per sweep (src/GibbsRtIrtCross.pl.jl:65,296); Post.mean.nu is its
    post-burn-in average reshaped N x J (column-major vec)."""
    pkg = pu.ge.load_package()
    z, Y, logT, X, init = load_golden("crossqr")
    N, J = Y.shape
    Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=0, nIter=6, nChain=1, qRt=0.85)
    M = pkg.GibbsRtIrtCrossQr(Cond, Data=pkg.InputData(Y=Y, T=np.exp(logT), X=np.zeros((N, 0))), precision="f64")
    pkg.sample_b(M)
    P = M.Post
    assert P.qr.shape == (6, J + 4 + N * J, 1)
    assert np.all(P.qr[:, J + 4:, 0] > 0)
    assert np.allclose(np.asarray(P.mean.nu).reshape(-1, order="F"), P.qr[3:, J + 4:, 0].mean(axis=0), rtol=1e-12)
    assert np.allclose(P.mean.rho, P.qr[3:, :J, 0].mean(axis=0))


@pytest.mark.parametrize("name,model,nchain", [("GibbsRtIrt", "rtirt", 1), ("GibbsRtIrt", "rtirt", 2), ("GibbsRtIrtLatentQr", "latentqr", 1),
                                               ("GibbsRtIrtCrossQr", "crossqr", 1), ("GibbsMlIrt", "mlirt", 2)])
def test_device_diagnostics_match_the_host_estimator(name, model, nchain):
    """erm_get_diagnostics (ESS and split-R-hat of every Post.ra / rt / qr column from the device-resident traces) against the numpy
    twin applied to the traces pulled to the host; checkConvergence's summary (src/SimTools.jl:419-443)."""
    pkg = pu.ge.load_package()
    z, Y, logT, X, init = load_golden(model)
    N, J = Y.shape
    Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=0 if model == "crossqr" else 3, nIter=120, nChain=nchain, qRt=0.85)
    D = pkg.InputData(Y=Y, T=np.exp(logT) if logT is not None else np.ones_like(Y, dtype=float), X=X if X is not None else np.zeros((N, 0)))
    M = getattr(pkg, name)(Cond, Data=D, precision="f32")
    pkg.sample_b(M)
    conv = pkg.checkConvergence(M)
    nb = Cond.nBurnin
    for tr_name in conv["detail"]:
        ess, rhat = conv["detail"][tr_name]
        tr = getattr(M.Post, tr_name)
        assert ess.shape == (tr.shape[1],)
        for k in list(range(0, tr.shape[1], max(1, tr.shape[1] // 40))) + [tr.shape[1] - 1]:
            e, r = pkg.ess_rhat(tr[nb:, k, :])
            if np.isnan(e):
                assert np.isnan(ess[k]) and np.isnan(rhat[k]), (tr_name, k)
            else:
                assert abs(ess[k] - e) <= 1e-6 * abs(e) and abs(rhat[k] - r) <= 1e-9, (tr_name, k, ess[k], e)
    assert 0 <= conv["ess"] <= 100 and 0 <= conv["rhat"] <= 100 and " / " in conv["essN"]
    assert conv["rhat"] > 50          # most subject-level parameters mix well within 60 post-burn-in iterations per chain


def test_run_simulation_like_the_reference():
    """runSimulation (src/SimTools.jl:457-495): nRep data sets from one truth, sample! on each, Post.mean / DIC / convergence summary
    per replication, then getMetrics over the replications."""
    pkg = pu.ge.load_package()
    Cond = pkg.setCond(nSubj=800, nItem=10, nFeat=2, nIter=300, nChain=1, nRep=3)
    tp = pkg.setTrueParaRtIrt(Cond, seed=3)
    Run = pkg.runSimulation(Cond, tp, Para=("a", "b", "λ", "σ²t"), funcData=pkg.setDataRtIrt, funcGibbs=pkg.GibbsRtIrt)
    assert set(Run) == {"True", 1, 2, 3} and set(Run[1]) == {"a", "b", "λ", "σ²t", "Dic", "Diag", "Seconds"}
    assert np.isfinite(Run[2]["Dic"][0]) and " / " in Run[3]["Diag"]["essN"]
    assert not np.array_equal(Run[1]["a"], Run[2]["a"])                      # different data sets
    m = pkg.getMetrics(Run, par="b")
    assert m["Rmse"] < 0.2 and m["Corr"] > 0.9 and abs(m["Bias"]) < 0.1
    null = pkg.runSimulation(pkg.setCond(nSubj=500, nItem=8, nFeat=0, nIter=200, nChain=1, nRep=1), tp.__class__(
        a=tp.a[:8], b=tp.b[:8], lam=tp.lam[:8], sig2t=tp.sig2t[:8], Sigp=np.eye(2)), funcData=pkg.setDataRtIrtNull, funcGibbs=pkg.GibbsRtIrtNull)
    assert pkg.getMetrics(null, par="a")["Rmse"] < 0.3
