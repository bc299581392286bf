"""The reference's own demo data set (data/demo.csv, read by its `test SimTools.jl`:186-203 as Y = columns 2:11, T = exp.(columns
12:21), X = columns 22:25) through every sampler: the HIP engine (fp64) against the oracle, on the configuration of that script
(nItem = 10, nFeat = 4, nChain = 3, GibbsRtIrtNull with cov2one = true) and on the other six samplers.  tests/golden/demo.csv is a
verbatim copy of the reference's DATA file (a fixture, not source)."""
import os

import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def load_demo():
    rows = np.genfromtxt(os.path.join(HERE, "golden", "demo.csv"), delimiter=",", skip_header=1, usecols=range(1, 25))
    Y = rows[:, 0:10].astype(np.uint8)
    logT = rows[:, 10:20]                       # T = exp.(Demo[:, 12:21]): the file holds log response times
    X = rows[:, 20:24]
    return Y, logT, X


def _init(model, N, J, F, g):
    st = dict(theta=g.standard_normal(N))
    if model != "mlirt":
        st.update(zeta=g.standard_normal(N), sigp=np.eye(2))
    if model == "mlirt":
        st["beta"] = g.standard_normal(F + 1)
    elif model == "rtirt":
        st["beta"] = g.standard_normal((F + 1, 2))
    elif model in ("latentqr", "latent"):
        st["beta"] = g.standard_normal(F + 2)
    elif model in ("crossqr", "cross"):
        st["rho"] = g.standard_normal(J)
    return st


@pytest.mark.parametrize("model", ["null", "rtirt", "mlirt", "latentqr", "latent", "crossqr", "cross"])
def test_demo_data_f64_parity(model):
    Y, logT, X = load_demo()
    N, J = Y.shape
    assert (N, J, X.shape[1]) == (300, 10, 4) and set(np.unique(Y)) == {0, 1} and logT.min() > 0
    Xm = None if model in ("crossqr", "cross") else X
    F = 0 if Xm is None else 4
    init = _init(model, N, J, F, np.random.default_rng(2))
    T = 3 if model == "crossqr" else 9                     # nChain = 3 interleaved pseudo-chains x 3 iterations (CrossQr: see test_gpu_parity)
    cov2one = model not in ("latentqr", "latent")
    dev = pu.run_device(model, Y, None if model == "mlirt" else logT, Xm, init, T, precision="f64", qRt=0.5, cov2one=cov2one,
                        n_chain=3 if model != "crossqr" else 1, n_burnin=1)
    op = pu.OracleProblem(model, Y, None if model == "mlirt" else logT, Xm, init, qRt=0.5, cov2one=cov2one)
    tr = op.run(T, with_nu=model in ("latentqr", "crossqr"))
    nC = 3 if model != "crossqr" else 1

    def rows(a):                                          # Julia layout (nIter, width, nChain) -> trace rows m * nChain + l
        return np.stack([a[r // nC, :, r % nC] for r in range(T)])

    assert pu.rel_err(rows(dev["ra"]), tr["ra"]).max() < 1e-8
    if model != "mlirt":
        assert pu.rel_err(rows(dev["rt"]), tr["rt"]).max() < 1e-8
    assert pu.rel_err(rows(dev["qr"]), tr["qr"]).max() < 1e-8
    assert pu.rel_err(rows(dev["ll"])[:, 0], tr["ll"]).max() < 1e-9


def test_demo_data_null_model_like_the_reference_script():
    """`MCMC7 = GibbsRtIrtNull(Cond7_50, Data = Data7); sample!(MCMC7, cov2one = true); coef(MCMC7)` (test SimTools.jl:194-203) with a
    shorter chain: finite posterior means in plausible ranges, improving log-likelihood, unit-diagonal Sigma_p."""
    pkg = pu.ge.load_package()
    Y, logT, X = load_demo()
    Cond = pkg.setCond(nSubj=Y.shape[0], nItem=10, nFeat=4, nChain=3, nIter=300, nThin=3, qRt=0.5)
    M = pkg.GibbsRtIrtNull(Cond, Data=pkg.InputData(Y=Y, T=np.exp(logT), X=X))
    pkg.sample_b(M, cov2one=True)
    c = pkg.coef(M)
    assert np.all(np.isfinite(c["a"])) and np.all(c["a"] > 0) and np.all(np.abs(c["b"]) <= 4) and np.all(c["σ²t"] > 0)
    assert np.allclose(np.diag(c["Σp"]), 1.0) and abs(c["Σp"][0, 1]) < 1 and np.all(c["β"] == 0)
    assert 2.0 < c["λ"].mean() < 4.5                       # the demo's log response times average ~3
    ll = M.Post.logLike[:, 0, :]
    assert ll[-50:].mean() > ll[:3].mean()


def load_timss():
    """data/timms2019math.csv of the reference: 14 scored items (ME62*), their screen times in seconds (*_S) and ten standardised
    background scales (*_Z) of 631 students -- complete cases, no missing values."""
    import csv
    with open(os.path.join(HERE, "golden", "timms2019math.csv")) as f:
        rows = list(csv.reader(f))
    h = rows[0]
    a = np.array(rows[1:], dtype=np.float64)
    iy = [k for k, n in enumerate(h) if n.startswith("ME62") and not n.endswith("_S")]
    it = [k for k, n in enumerate(h) if n.startswith("ME62") and n.endswith("_S")]
    ix = [k for k, n in enumerate(h) if n.endswith("_Z")]
    return a[:, iy].astype(np.uint8), np.log(a[:, it]), a[:, ix]


@pytest.mark.parametrize("model", ["rtirt", "mlirt", "latentqr", "crossqr", "null"])
def test_timss_data_f64_parity(model):
    """A real assessment data set (14 items, 10 covariates, response times from 1.7 s to 19 min) through the engine and the oracle."""
    Y, logT, X = load_timss()
    N, J = Y.shape
    assert (N, J, X.shape[1]) == (631, 14, 10) and set(np.unique(Y)) == {0, 1} and np.all(np.isfinite(logT))
    Xm = None if model == "crossqr" else X
    F = 0 if Xm is None else 10
    init = _init(model, N, J, F, np.random.default_rng(3))
    T = 3 if model == "crossqr" else 8
    cov2one = model != "latentqr"
    res_dev = pu.run_device(model, Y, None if model == "mlirt" else logT, Xm, init, T, precision="f64", qRt=0.85, cov2one=cov2one)
    tr = pu.OracleProblem(model, Y, None if model == "mlirt" else logT, Xm, init, qRt=0.85, cov2one=cov2one).run(T, with_nu=model in ("latentqr", "crossqr"))
    assert pu.rel_err(res_dev["ra"][:, :, 0], tr["ra"]).max() < 1e-8
    if model != "mlirt":
        assert pu.rel_err(res_dev["rt"][:, :, 0], tr["rt"]).max() < 1e-8
    assert pu.rel_err(res_dev["qr"][:, :, 0], tr["qr"]).max() < 1e-8
    assert pu.rel_err(res_dev["ll"][:, 0, 0], tr["ll"]).max() < 1e-9


def test_timss_quantile_fit_like_the_readme():
    """README.md:84-102: Cond = setCond(qRa=0.85, qRt=0.85, nChain=3, nIter=...); MCMC = GibbsRtIrtQuantile(Cond, Data=Data); sample!;
    coef; Post.mean.Sigp / beta -- on the reference's TIMSS file, fp32 engine."""
    pkg = pu.ge.load_package()
    Y, logT, X = load_timss()
    Cond = pkg.setCond(nSubj=631, nItem=14, nFeat=10, qRa=0.85, qRt=0.85, nChain=3, nIter=200)
    M = pkg.GibbsRtIrtQuantile(Cond, Data=pkg.InputData(Y=Y, T=np.exp(logT), X=X))
    pkg.sample_b(M)
    c = pkg.coef(M)
    assert np.all(np.isfinite(c["a"])) and np.all(c["a"] > 0) and np.all(np.isfinite(c["β"])) and c["β"].shape == (12,) and c["β"][0] == 0
    assert np.asarray(M.Post.mean.Sigp).shape == (4,) and M.Post.mean.Sigp[0] == 1 and M.Post.mean.Sigp[3] > 0
    assert M.Post.qr.shape == (200, 10 + 2 + 4 + 631, 3)
    assert abs(c["λ"].mean() - logT.mean()) < 1.0 and np.all(np.isfinite(M.Post.logLike))
