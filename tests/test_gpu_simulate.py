"""Synthetic data generated on the device (erm_simulate_data; setData* of src/SimTools.jl:117-368): distributional checks of every
generator, equivalence of the installed data set with erm_set_data on the same values, and parameter recovery end to end."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu
pkg = pu.ge.load_package()
L = pkg._lib


def _cond(N=4000, J=10, F=2, nIter=240):
    return pkg.setCond(nSubj=N, nItem=J, nFeat=F, nIter=nIter, nChain=1, qRt=0.5)


@pytest.mark.parametrize("name,truth,gen", [("GibbsMlIrt", "setTrueParaMlIrt", "mlirt"), ("GibbsRtIrt", "setTrueParaRtIrt", "rtirt"),
                                            ("GibbsRtIrtNull", "setTrueParaRtIrt", "null"), ("GibbsRtIrtCross", "setTrueParaRtIrtCross", "cross"),
                                            ("GibbsRtIrtLatent", "setTrueParaRtIrtLatent", "latent")])
def test_generators_follow_the_reference_distributions(name, truth, gen):
    C = _cond()
    N, J, F = C.nSubj, C.nItem, C.nFeat
    tp = getattr(pkg, truth)(C, seed=5)
    M = getattr(pkg, name)(C, precision="f64")
    pkg.simulateData(M, tp, seed=77)
    D, th, ze = M.Data, tp.theta, tp.zeta
    Y = np.asarray(D.Y, dtype=float)
    pr = 1 / (1 + np.exp(-tp.a[None, :] * (th[:, None] - tp.b[None, :])))
    assert np.max(np.abs(Y.mean(0) - pr.mean(0))) < 4 * 0.5 / np.sqrt(N)                   # Bernoulli(logistic(a (theta - b)))
    assert abs(np.corrcoef((Y - pr).ravel(), pr.ravel())[0, 1]) < 0.02
    if gen in ("mlirt", "rtirt", "latent"):
        X = np.asarray(D.X)
        assert X.shape == (N, F) and abs(X[:, 1].mean()) < 0.08 and abs(X[:, 1].std() - 1) < 0.05
    if gen == "mlirt":
        assert set(np.unique(np.asarray(D.X)[:, 0])) == {0.0, 1.0}                          # X[:,1] ~ Bernoulli(0.5)  (src/SimTools.jl:353)
        r = th - np.asarray(D.X) @ np.asarray(tp.beta).ravel()
        assert abs(r.mean()) < 0.06 and abs(r.std() - 1) < 0.05
        return
    logT = np.asarray(D.logT)
    if gen == "rtirt":
        B = np.asarray(tp.beta).reshape(F, 2)
        r = np.column_stack([th, ze]) - np.asarray(D.X) @ B
        assert np.max(np.abs(np.cov(r.T) - np.eye(2))) < 0.08
    if gen in ("null", "cross"):
        assert np.max(np.abs(np.cov(np.column_stack([th, ze]).T) - np.eye(2))) < 0.08
    if gen in ("rtirt", "null"):
        assert logT.min() > 0                                                                # truncated at 0 (src/SimTools.jl:169)
        z = (logT - (tp.lam[None, :] - ze[:, None])) / np.sqrt(tp.sig2t)[None, :]
        ok = (tp.lam[None, :] - ze[:, None]) / np.sqrt(tp.sig2t)[None, :] > 4                # cells where the truncation is immaterial
        assert abs(z[ok].mean()) < 0.02 and abs(z[ok].std() - 1) < 0.02
    if gen == "cross":
        e = (logT - (tp.lam[None, :] - ze[:, None] - th[:, None] * tp.rho[None, :])) / 0.3
        assert abs(e.mean()) < 0.02 and abs(e.std() - 1) < 0.02
    if gen == "latent":
        x = np.column_stack([np.asarray(D.X), th])
        e = (ze - x @ np.asarray(tp.beta)) / 0.3
        assert abs(th.std() - 1) < 0.05 and abs(e.mean()) < 0.05 and abs(e.std() - 1) < 0.05
        assert abs((logT - (tp.lam[None, :] - ze[:, None])).std() - 1) < 0.02


@pytest.mark.parametrize("type_,check", [("tail", lambda e: 1.2 < e.std() < 1.4 and np.mean(np.abs(e) > 3) > 0.015),   # t5: sd sqrt(5/3), heavy tails
                                         ("skew", lambda e: abs(e.mean() + 0.5) < 0.02 and abs(e.std() - np.sqrt(0.5)) < 0.03 and e.min() >= -1)])
def test_noise_types_of_the_cross_generator(type_, check):
    """rand(TDist(5)) / rand(Gamma(0.5, 1)) .- 1 enter logT UNSCALED (src/SimTools.jl:240-245); only the normal type has sd 0.3."""
    C = _cond(N=6000, J=8, F=0)
    tp = pkg.setTrueParaRtIrtCross(C, seed=2)
    M = pkg.GibbsRtIrtCross(C, precision="f64")
    pkg.simulateData(M, tp, type=type_, seed=9)
    e = np.asarray(M.Data.logT) - (tp.lam[None, :] - tp.zeta[:, None] - tp.theta[:, None] * tp.rho[None, :])
    assert check(e.ravel())


GENS = [("GibbsMlIrt", "setTrueParaMlIrt", 0), ("GibbsRtIrt", "setTrueParaRtIrt", 1), ("GibbsRtIrtNull", "setTrueParaRtIrt", 2),
        ("GibbsRtIrtCross", "setTrueParaRtIrtCross", 3), ("GibbsRtIrtLatent", "setTrueParaRtIrtLatent", 4)]


@pytest.mark.parametrize("name,truth,gen", GENS)
@pytest.mark.parametrize("type_", ["norm", "tail", "skew"])
def test_device_generators_equal_the_oracle_restatement_value_by_value(name, truth, gen, type_):
    """erm_simulate_data against oracle/erm_oracle.c::orc_simulate_data (setData* of src/SimTools.jl:117-368 restated with the same
    (DATA_SUBJ, i) / (DATA_CELL, i, j) stream addressing): X, theta, zeta, logT to 1e-12, Y bit for bit (fp64 engine)."""
    if type_ != "norm" and gen not in (3, 4):
        pytest.skip("only setDataRtIrtCross / setDataRtIrtLatent take a noise type")
    C = _cond(N=3000, J=9, F=3)
    N, J, F = C.nSubj, C.nItem, C.nFeat
    tp = getattr(pkg, truth)(C, seed=11)
    if gen == 1:
        tp.Sigp = np.array([[1.3, 0.4], [0.4, 0.8]])               # a full covariance exercises the Cholesky factor
    M = getattr(pkg, name)(C, precision="f64")
    pkg.simulateData(M, tp, type=type_, seed=2024)
    Fk = 0 if gen in (2, 3) else F
    o = pu.orc_simulate(gen, N, J, Fk, a=tp.a, b=tp.b, lam=tp.lam if gen else None, sig2t=(tp.sig2t if np.size(tp.sig2t) else np.ones(J)) if gen else None,
                        rho=tp.rho if gen == 3 else None, Sigp=np.asarray(tp.Sigp).reshape(-1, order="F") if np.size(tp.Sigp) else None,
                        beta=np.asarray(tp.beta).reshape(-1, order="F") if Fk else None, seed=2024, noise={"norm": 0, "tail": 1, "skew": 2}[type_])
    assert np.array_equal(np.asarray(M.Data.Y, dtype=np.uint8), o["Y"])
    assert np.max(np.abs(tp.theta - o["theta"])) < 1e-12
    if gen:
        assert np.max(np.abs(tp.zeta - o["zeta"])) < 1e-12
        assert np.max(np.abs(np.asarray(M.Data.logT) - o["logT"])) < 1e-12
    if Fk:
        assert np.max(np.abs(np.asarray(M.Data.X) - o["X"])) < 1e-12



def test_installed_data_set_equals_set_data_on_the_same_values():
    """The constants erm_simulate_data derives on the device (K0, column means and centred squares of logT, x'x and its inverse) must
    be those erm_set_data derives on the host from the same values: identical chains (fp64 engine, 1e-9)."""
    C = _cond(N=1500, J=9, F=3, nIter=8)
    tp = pkg.setTrueParaRtIrt(C, seed=4)
    M = pkg.GibbsRtIrt(C, precision="f64", seed=42)
    pkg.simulateData(M, tp, seed=1)
    init = M._state_for_engine()
    pkg.sample_b(M)
    M2 = pkg.GibbsRtIrt(C, Data=M.Data, precision="f64", seed=42)
    M2.Para = M2.Para.__class__(**{k: np.array(v, copy=True) for k, v in dict(theta=init["theta"], a=init["a"], b=init["b"], zeta=init["zeta"],
                                lam=init["lambda_"], sig2t=init["sig2t"], beta=init["beta"].reshape(4, 2, order="F"), Sigp=init["sigp"].reshape(2, 2)).items()})
    pkg.sample_b(M2)
    assert pu.rel_err(M.Post.ra, M2.Post.ra).max() < 1e-9 and pu.rel_err(M.Post.rt, M2.Post.rt).max() < 1e-9
    assert pu.rel_err(M.Post.logLike, M2.Post.logLike).max() < 1e-10


def test_simulate_then_sample_recovers_the_truth_without_touching_the_host():
    C = _cond(N=5000, J=12, F=2, nIter=300)
    tp = pkg.setTrueParaRtIrt(C, seed=8)
    M = pkg.GibbsRtIrt(C, precision="f32", trace="summary")
    pkg.simulateData(M, tp, seed=3, pull=False)
    assert M.Data is None
    pkg.sample_b(M)
    P = M.Post.mean
    assert pkg.getRmse(P.a, tp.a) < 0.1 and pkg.getRmse(P.b, tp.b) < 0.1 and np.corrcoef(P.theta, tp.theta)[0, 1] > 0.85
    assert np.corrcoef(P.zeta, tp.zeta)[0, 1] > 0.98
    pkg.sample_b(M, itemtype="1pl")                               # a new engine (other kwargs) inherits the device-resident data set
    assert np.all(M.Post.mean.a == 1)
