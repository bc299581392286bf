"""Build-owned pins of the oracle's restatement of the reference's data generators (oracle/erm_oracle.c::orc_simulate_data; setData* of
/root/reference/src/SimTools.jl:117-368): every generated quantity has the distribution the reference draws it from.  The device
generators are then compared with this restatement value by value (tests/test_gpu_simulate.py)."""
import numpy as np
import pytest

import parity_util as pu

N, J, F = 20000, 6, 3
g = np.random.default_rng(0)
a, b = np.abs(g.normal(1, 0.2, J)), g.normal(0, 0.5, J)
lam, sig2t, rho = g.normal(4, 0.2, J), np.exp(g.normal(np.log(0.3), 0.2, J)), g.normal(0, 0.2, J)


def _bernoulli_ok(Y, th):
    pr = 1 / (1 + np.exp(-a[None, :] * (th[:, None] - b[None, :])))
    return np.max(np.abs(Y.mean(0) - pr.mean(0))) < 4 * 0.5 / np.sqrt(N) and abs(np.corrcoef((Y - pr).ravel(), pr.ravel())[0, 1]) < 0.02


def test_mlirt_generator():                      # src/SimTools.jl:349-368
    beta = np.array([0.7, -0.4, 0.2])
    o = pu.orc_simulate(0, N, J, F, a=a, b=b, beta=beta)
    X = o["X"]
    assert set(np.unique(X[:, 0])) == {0.0, 1.0} and abs(X[:, 0].mean() - 0.5) < 0.02            # Bernoulli(0.5)
    assert abs(X[:, 1].mean()) < 0.03 and abs(X[:, 1].std() - 1) < 0.03
    r = o["theta"] - X @ beta
    assert abs(r.mean()) < 0.03 and abs(r.std() - 1) < 0.03                                      # Normal(X beta, 1)
    assert _bernoulli_ok(o["Y"].astype(float), o["theta"])


@pytest.mark.parametrize("gen", [1, 2])
def test_rtirt_and_null_generators(gen):         # src/SimTools.jl:149-178, 117-144
    Sigp = np.array([[1.3, 0.4], [0.4, 0.8]])
    beta = g.normal(0, 1, (F, 2))
    o = pu.orc_simulate(gen, N, J, F if gen == 1 else 0, a=a, b=b, lam=lam, sig2t=sig2t, Sigp=Sigp.reshape(-1, order="F"), beta=beta if gen == 1 else None)
    sub = np.column_stack([o["theta"], o["zeta"]])
    if gen == 1:
        sub = sub - o["X"] @ beta
    assert np.max(np.abs(np.cov(sub.T) - Sigp)) < 0.05 and np.max(np.abs(sub.mean(0))) < 0.03  # MvNormal(0, Sigp)
    assert _bernoulli_ok(o["Y"].astype(float), o["theta"])
    logT, mu = o["logT"], lam[None, :] - o["zeta"][:, None]
    assert logT.min() > 0                                                                       # Truncated(Normal(mu, sqrt(sig2t)), 0, Inf)
    z = (logT - mu) / np.sqrt(sig2t)[None, :]
    far = mu / np.sqrt(sig2t)[None, :] > 4
    assert abs(z[far].mean()) < 0.02 and abs(z[far].std() - 1) < 0.02


@pytest.mark.parametrize("noise,check", [(0, lambda e: abs(e.mean()) < 0.01 and abs(e.std() - 0.3) < 0.01),
                                         (1, lambda e: abs(e.mean()) < 0.02 and 1.2 < e.std() < 1.4 and np.mean(np.abs(e) > 3) > 0.015),
                                         (2, lambda e: abs(e.mean() + 0.5) < 0.02 and abs(e.std() - np.sqrt(0.5)) < 0.03 and e.min() >= -1)])
def test_cross_and_latent_generators(noise, check):      # src/SimTools.jl:220-255, 304-343
    o = pu.orc_simulate(3, N, J, 0, a=a, b=b, lam=lam, sig2t=sig2t, rho=rho, noise=noise)
    assert np.max(np.abs(np.cov(np.column_stack([o["theta"], o["zeta"]]).T) - np.eye(2))) < 0.05
    e = o["logT"] - (lam[None, :] - o["zeta"][:, None] - o["theta"][:, None] * rho[None, :])
    assert check(e.ravel()) and _bernoulli_ok(o["Y"].astype(float), o["theta"])
    beta = np.array([0.3, -0.2, 0.5, 0.4])
    o = pu.orc_simulate(4, N, J, F, a=a, b=b, lam=lam, sig2t=sig2t, beta=beta, noise=noise)
    e = o["zeta"] - np.column_stack([o["X"], o["theta"]]) @ beta
    assert check(e) and abs(o["theta"].std() - 1) < 0.03
    r = o["logT"] - (lam[None, :] - o["zeta"][:, None])
    assert abs(r.mean()) < 0.01 and abs(r.std() - 1) < 0.01                                     # + randn


def test_generators_are_addressed_not_sequenced():
    """Subject i's values depend on (seed, i) only: a shorter data set is a prefix of a longer one, and the thread count is immaterial."""
    big = pu.orc_simulate(1, 500, J, F, a=a, b=b, lam=lam, sig2t=sig2t, beta=np.ones((F, 2)))
    small = pu.orc_simulate(1, 200, J, F, a=a, b=b, lam=lam, sig2t=sig2t, beta=np.ones((F, 2)))
    assert np.array_equal(big["logT"][:200], small["logT"]) and np.array_equal(big["Y"][:200], small["Y"]) and np.array_equal(big["X"][:200], small["X"])
    pu.oracle().orc_set_threads(4)
    try:
        par = pu.orc_simulate(1, 500, J, F, a=a, b=b, lam=lam, sig2t=sig2t, beta=np.ones((F, 2)))
    finally:
        pu.oracle().orc_set_threads(1)
    assert all(np.array_equal(big[k], par[k]) for k in big)
