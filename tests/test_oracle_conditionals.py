"""The oracle's full conditionals against numpy re-evaluations of the reference's broadcast expressions
(/root/reference/src/Draw.pl.jl; numpy broadcasting mirrors Julia's dot-broadcasting line by line)."""
import numpy as np
import pytest

import parity_util as pu

N, J, F = 211, 9, 3


def _problem(model, seed=0):
    Y, logT, X, init, _ = pu.make_problem(model, N, J, F, seed=seed)
    g = np.random.default_rng(seed + 100)
    st = dict(init)
    st.update(a=np.exp(g.normal(0, 0.3, J)), b=g.normal(0, 1, J))
    if model == "cross":
        st.update(rho=g.normal(0, 0.3, J))
    if model != "mlirt":
        st.update(lam=g.normal(3, 0.5, J), sig2t=np.exp(g.normal(-1, 0.3, J)), sigp=np.array([[1.3, 0.2], [0.2, 0.8]]))
    if model == "crossqr":
        st.update(nu=np.exp(g.normal(0, 0.7, (N, J))))
    if model == "latentqr":
        st.update(nu=np.exp(g.normal(0, 0.7, N)))
    op = pu.OracleProblem(model, Y, logT, X, st, qRt=0.85)
    op.arr["omega"][:] = np.exp(g.normal(-1.5, 0.3, N * J))
    return op, Y, logT, X, st


def _P(op, Y, logT, X):
    a = op.arr
    P = dict(th=a["theta"][:, None], ze=a["zeta"][:, None], a=a["a"][None, :], b=a["b"][None, :], lam=a["lambda_"][None, :],
             s2=a["sig2t"][None, :], rho=a["rho"][None, :], om=a["omega"].reshape(N, J, order="F"), kap=Y.astype(float) - 0.5,
             logT=logT, Sigp=a["Sigp"].reshape(2, 2, order="F"))
    if X is not None:
        P["x"] = np.column_stack([np.ones(N), X])
    return P


def test_theta_moments():          # src/Draw.pl.jl:49-62
    op, Y, logT, X, st = _problem("rtirt")
    P = _P(op, Y, logT, X)
    mu0 = P["x"] @ op.arr["beta"].reshape(F + 1, 2, order="F")[:, 0]
    s0 = P["Sigp"][0, 0]
    parV = 1 / (1 / s0 + np.sum(P["a"] ** 2 * P["om"], axis=1))
    parM = parV * (mu0 / s0 + np.sum(P["a"] * (P["kap"] + P["a"] * P["b"] * P["om"]), axis=1))
    m, v = op.moments(0, N, N)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    m0, v0 = op.moments(1, N, N)     # Null prior, :67-80
    assert np.allclose(m0, parV * np.sum(P["a"] * (P["kap"] + P["a"] * P["b"] * P["om"]), axis=1), rtol=1e-12)


def test_item_moments():           # a :88-93, b :98-105
    op, Y, logT, X, st = _problem("rtirt")
    P = _P(op, Y, logT, X)
    parV = 1 / (1 + np.sum((P["th"] - P["b"]) ** 2 * P["om"], axis=0))
    parM = parV * (1 + np.sum(P["kap"] * (P["th"] - P["b"]), axis=0))
    m, v = op.moments(2, J, J)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    parV = 1 / (1 + np.sum(P["a"] ** 2 * P["om"], axis=0))
    parM = parV * (0 - np.sum(P["a"] * (P["kap"] - (P["th"] * P["a"]) * P["om"]), axis=0))
    m, v = op.moments(3, J, J)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)


def test_speed_intensity_residual_moments():    # zeta :132-141, lambda :215-220, sig2t :257-262
    op, Y, logT, X, st = _problem("rtirt")
    P = _P(op, Y, logT, X)
    mu0 = P["x"] @ op.arr["beta"].reshape(F + 1, 2, order="F")[:, 1]
    s0 = P["Sigp"][1, 1]
    parV = 1 / (1 / s0 + np.sum(1 / P["s2"], axis=1))
    parM = parV * (mu0 / s0 + np.sum((P["lam"] - logT) / P["s2"], axis=1))
    m, v = op.moments(4, N, N)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    mu, sd = logT.mean(), logT.std(ddof=1)
    parV = 1 / (1 / sd ** 2 + N / P["s2"][0])
    parM = parV * (mu / sd ** 2 + np.sum(logT + P["ze"], axis=0) / P["s2"][0])
    m, v = op.moments(5, J, J)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    sh, sc = op.moments(6, J, J)
    assert np.allclose(sh, 1e-3 + N / 2) and np.allclose(sc, 1e-3 + np.sum((logT - P["lam"] + P["ze"]) ** 2, axis=0) / 2, rtol=1e-12)


def test_beta_and_sigp_rtirt():    # drawSubjCoefficients :380-393 (1 added to EVERY element), drawSubjCovariance :499-505
    op, Y, logT, X, st = _problem("rtirt")
    P = _P(op, Y, logT, X)
    x, eta = P["x"], np.column_stack([op.arr["theta"], op.arr["zeta"]])
    iO = np.linalg.inv(P["Sigp"])
    parV = np.linalg.inv(1.0 + np.kron(iO, x.T @ x))
    parM = parV @ (0.0 + (x.T @ eta @ iO.T).reshape(-1, order="F"))
    n = 2 * (F + 1)
    m, v = op.moments(8, n, n * n)
    assert np.allclose(m, parM, rtol=1e-9) and np.allclose(v.reshape(n, n, order="F"), parV, rtol=1e-9)
    e = eta - x @ op.arr["beta"].reshape(F + 1, 2, order="F")
    psi, _ = op.moments(9, 4)
    assert np.allclose(psi.reshape(2, 2, order="F"), e.T @ e + np.eye(2), rtol=1e-12)


def test_crossqr_moments():        # zeta :192-206, lambda :239-251, sig2t :278-288, rho :474-489
    op, Y, logT, X, st = _problem("crossqr")
    P = _P(op, Y, logT, None)
    q = 0.85
    k1, k2 = (1 - 2 * q) / (q * (1 - q)), 2 / (q * (1 - q))
    nu = op.arr["nu"].reshape(N, J, order="F")
    k1e, k2e = k1 * nu, k2 * nu
    s0 = P["Sigp"][1, 1]
    parV = 1 / (1 / s0 + np.sum(1 / (P["s2"] * k2e), axis=1))
    parM = parV * np.sum((P["lam"] - logT - P["th"] * P["rho"] + k1e) / (P["s2"] * k2e), axis=1)
    m, v = op.moments(4, N, N)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    mu, sd = logT.mean(), logT.std(ddof=1)
    parV = 1 / (1 / sd ** 2 + np.sum(1 / (P["s2"] * k2e), axis=0))
    parM = parV * (mu / sd ** 2 + np.sum((logT + P["ze"] + P["th"] * P["rho"] - k1e) / (P["s2"] * k2e), axis=0))
    m, v = op.moments(5, J, J)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    sh, sc = op.moments(6, J, J)
    want = 1e-3 + np.sum((logT - P["lam"] + P["ze"] + P["th"] * P["rho"] - k1e) ** 2 / (2 * k2e), axis=0) + nu.sum(0)
    assert np.allclose(sh, 1e-3 + 3 * N / 2) and np.allclose(sc, want, rtol=1e-12)
    parV = 1 / (1 + np.sum(P["th"] ** 2 / (P["s2"] * k2e), axis=0))
    parM = parV * np.sum(P["th"] * (P["lam"] - P["ze"] - logT + k1e) / (P["s2"] * k2e), axis=0)
    m, v = op.moments(7, J, J)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)


def test_latentqr_moments_and_matrix_division_quirk():
    """zeta :161-174; Sigma_p scale :594 where `r.^2 / (2*k2e)` is a vector/vector division: the N x N matrix
    r2 w'/(w'w), whose sum the oracle evaluates in closed form."""
    op, Y, logT, X, st = _problem("latentqr")
    P = _P(op, Y, logT, X)
    q = 0.85
    k1, k2 = (1 - 2 * q) / (q * (1 - q)), 2 / (q * (1 - q))
    nu = op.arr["nu"]
    x = np.column_stack([np.ones(N), X, op.arr["theta"]])
    mu0 = x @ op.arr["beta"] + k1 * nu
    s0 = P["Sigp"][1, 1] * (k2 * nu)
    parV = 1 / (1 / s0 + np.sum(1 / P["s2"], axis=1))
    parM = parV * (mu0 / s0 + np.sum((P["lam"] - logT) / P["s2"], axis=1))
    m, v = op.moments(4, N, N)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    r2 = (op.arr["zeta"] - x @ op.arr["beta"] - k1 * nu) ** 2
    w = 2 * k2 * nu
    matrix = np.outer(r2, w) / (w @ w)            # Julia: a / b == a * pinv(b) for vectors
    sc, _ = op.moments(10, 1)
    assert np.allclose(sc[0], 1e-3 + matrix.sum() + nu.sum(), rtol=1e-11)


def test_beta_latentqr_is_the_stacked_least_squares_solution():   # getSubjCoefficientsLatentQr :446-458
    op, Y, logT, X, st = _problem("latentqr")
    q = 0.85
    k1, k2 = (1 - 2 * q) / (q * (1 - q)), 2 / (q * (1 - q))
    nu = op.arr["nu"].copy()
    x = np.column_stack([np.ones(N), X, op.arr["theta"]])
    w = 1 / (op.arr["Sigp"][3] * k2 * nu)
    Abig = np.kron(w[:, None], x.T @ x)                                   # (N p) x p
    rhs = (np.outer(x.T @ (op.arr["zeta"] - k1 * nu), w)).reshape(-1, order="F")
    want = np.linalg.lstsq(Abig, rhs, rcond=None)[0]
    want[0] = 0.0
    op.step(13, 1)
    assert np.allclose(op.arr["beta"], want, rtol=1e-8, atol=1e-10)


def test_null_cross_latent_variant_moments():
    """The non-quantile variants: drawSubjSpeedNull :119-127, drawSubjSpeedLatent :147-156,
    drawSubjSpeedCross :179-187, drawItemIntensityCross :225-231, drawItemTimeResidualCross :267-273, drawSubjCorrCross :463-469,
    drawSubjCoefficientsLatent :399-416 (1 added to EVERY element again), drawSubjCovarianceLatent :563-579,
    drawSubjCovarianceNull :522-535."""
    # ---- Null: prior N(0, 1) for zeta -- Sigp[2,2] is NOT used
    op, Y, logT, X, st = _problem("null")
    P = _P(op, Y, logT, X)
    parV = 1 / (1 / 1.0 + np.sum(1 / P["s2"], axis=1))
    parM = parV * (0.0 / 1.0 + np.sum((P["lam"] - logT) / P["s2"], axis=1))
    m, v = op.moments(4, N, N)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    eta = np.column_stack([op.arr["theta"], op.arr["zeta"]])
    Psi, _ = op.moments(9, 4)
    assert np.allclose(Psi.reshape(2, 2), eta.T @ eta + np.eye(2), rtol=1e-12)
    # ---- Cross
    op, Y, logT, X, st = _problem("cross")
    P = _P(op, Y, logT, None)
    s0 = P["Sigp"][1, 1]
    parV = 1 / (1 / s0 + np.sum(1 / P["s2"], axis=1))
    parM = parV * (0.0 / s0 + np.sum((P["lam"] - logT - P["th"] * P["rho"]) / P["s2"], axis=1))
    m, v = op.moments(4, N, N)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    mu, sd = logT.mean(), logT.std(ddof=1)
    parV = 1 / (1 / sd ** 2 + np.sum(N / P["s2"], axis=0))
    parM = parV * (mu / sd ** 2 + np.sum((logT + P["ze"] + P["th"] * P["rho"]) / P["s2"], axis=0))
    m, v = op.moments(5, J, J)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    sh, sc = op.moments(6, J, J)
    assert np.allclose(sh, 1e-3 + N / 2) and np.allclose(sc, 1e-3 + np.sum((logT - P["lam"] + P["ze"] + P["th"] * P["rho"]) ** 2, axis=0) / 2, rtol=1e-12)
    parV = 1 / (1 + np.sum(P["th"] ** 2 / P["s2"], axis=0))
    parM = parV * (0.0 + np.sum(P["th"] * (P["lam"] - P["ze"] - logT) / P["s2"], axis=0))
    m, v = op.moments(7, J, J)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    # ---- Latent
    op, Y, logT, X, st = _problem("latent")
    P = _P(op, Y, logT, X)
    x = np.column_stack([np.ones(N), X, op.arr["theta"]])
    s0 = P["Sigp"][1, 1]
    parV = 1 / (1 / s0 + np.sum(1 / P["s2"], axis=1))
    parM = parV * ((x @ op.arr["beta"]) / s0 + np.sum((P["lam"] - logT) / P["s2"], axis=1))
    m, v = op.moments(4, N, N)
    assert np.allclose(m, parM, rtol=1e-12) and np.allclose(v, parV, rtol=1e-12)
    invO = 1 / s0
    bV = np.linalg.inv(1.0 + invO * (x.T @ x))
    bM = bV @ (0.0 + x.T @ op.arr["zeta"] * invO)
    m, v = op.moments(11, F + 2, (F + 2) ** 2)
    assert np.allclose(m, bM, rtol=1e-9) and np.allclose(v.reshape(F + 2, F + 2, order="F"), bV, rtol=1e-9)
    sc, _ = op.moments(12, 1)
    assert np.allclose(sc[0], 1e-3 + np.sum((op.arr["zeta"] - x @ op.arr["beta"]) ** 2) / 2, rtol=1e-12)


def test_loglik_matches_host_post_processing():
    """orc_loglik vs the package's getLogLikelihood (numpy transcription of the reference's getLogLikelihood* functions)."""
    pkg = pu.ge.load_package()
    for model, cls in (("mlirt", pkg.GibbsMlIrt), ("rtirt", pkg.GibbsRtIrt), ("crossqr", pkg.GibbsRtIrtCrossQr), ("latentqr", pkg.GibbsRtIrtLatentQr)):
        op, Y, logT, X, st = _problem(model)
        Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=F, nIter=4, nChain=1, qRt=0.85)
        D = pkg.InputData(Y=Y, T=np.exp(logT) if logT is not None else (), X=X if X is not None else ())
        M = cls.__new__(cls)
        M.Cond, M.Data = Cond, D
        a = op.arr
        Pm = pkg.InputPara(theta=a["theta"], a=a["a"], b=a["b"], zeta=a["zeta"], lam=a["lambda_"], sig2t=a["sig2t"], beta=a["beta"],
                           Sigp=a["Sigp"], rho=a["rho"], nu=a["nu"])
        assert np.isclose(op.loglik(), pkg.getLogLikelihood(M, Pm), rtol=1e-10)
