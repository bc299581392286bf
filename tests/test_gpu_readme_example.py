"""BASELINE.json configs[0] -- the README's own example (/root/reference/README.md:61-77): setCond(nSubj=1000, nItem=15),
setTrueParaMlIrt, setDataMlIrt, GibbsMlIrt, sample!, getRmse / getBias on b -- through the HIP path (Julia-surface mirror, C-ABI),
nIter = 500, nChain = 1, in fp64 and in fp32, against the oracle's 500-sweep chain from the same initial values.

Tolerance: posterior means within 3 Monte-Carlo standard errors, the standard error of each parameter taken from
the ORACLE chain: sd of its post-burn-in draws / sqrt(effective sample size)."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu


def mcse(draws):
    """Monte-Carlo standard error of the mean of each column of `draws` (iterations x parameters): sd / sqrt(ESS), ESS by the package's
    split-chain Geyer estimator (gibbs.ess_rhat, itself pinned by known-answer tests in tests/test_host_api.py)."""
    pkg = pu.ge.load_package()
    out = np.empty(draws.shape[1])
    for k in range(draws.shape[1]):
        ess, _ = pkg.gibbs.ess_rhat(draws[:, k])
        out[k] = draws[:, k].std(ddof=1) / np.sqrt(min(max(ess, 1.0), draws.shape[0]))
    return out


@pytest.fixture(scope="module")
def example():
    pkg = pu.ge.load_package()
    Cond = pkg.setCond(nSubj=1000, nItem=15, nIter=500, nChain=1)
    g = np.random.default_rng(2024)
    truePara = pkg.setTrueParaMlIrt(Cond, seed=g)
    Data = pkg.setDataMlIrt(Cond, truePara, seed=g)
    M0 = pkg.GibbsMlIrt(Cond, Data=Data, truePara=truePara, precision="f64")
    init = dict(theta=M0.Para.theta.copy(), beta=M0.Para.beta.copy())
    op = pu.OracleProblem("mlirt", Data.Y, None, Data.X, init, seed=M0.seed)
    tr = op.run(500)
    N, J = 1000, 15
    post = slice(Cond.nBurnin, 500)
    chains = dict(a=tr["ra"][post, N:N + J], b=tr["ra"][post, N + J:], beta=tr["qr"][post, 1:])      # beta[0] = 0 (intercept=false) never moves
    return pkg, Cond, Data, truePara, chains


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_readme_example_posterior_means_within_three_mc_standard_errors(example, precision):
    pkg, Cond, Data, truePara, chains = example
    MCMC = pkg.GibbsMlIrt(Cond, Data=Data, truePara=truePara, precision=precision)
    pkg.sample_b(MCMC)
    got = dict(a=MCMC.Post.mean.a, b=MCMC.Post.mean.b, beta=np.asarray(MCMC.Post.mean.beta).reshape(-1)[1:])
    for name, ch in chains.items():
        se = mcse(ch)
        z = np.abs(got[name] - ch.mean(0)) / se
        assert z.max() < 3.0, (name, precision, z.max())
        if precision == "f64":
            assert np.max(np.abs(got[name] - ch.mean(0))) < 1e-7          # same chain, elementwise
    # the README's own check
    assert pkg.getRmse(truePara.b, MCMC.Post.mean.b) < 0.15 and abs(pkg.getBias(truePara.b, MCMC.Post.mean.b)) < 0.1
    assert MCMC.Post.ra.shape == (500, 1000 + 30, 1) and MCMC.Post.qr.shape == (500, Cond.nFeat + 1, 1)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_readme_example_independent_chain_agrees_within_mc_error(example, precision):
    """A device chain on ANOTHER random stream (seed) is an independent estimate of the same posterior means: the difference of two
    independent estimates has standard error sqrt(2) x MCSE; 4 such standard errors over the 33 monitored parameters."""
    pkg, Cond, Data, truePara, chains = example
    MCMC = pkg.GibbsMlIrt(Cond, Data=Data, truePara=truePara, precision=precision, seed=777)
    pkg.sample_b(MCMC)
    got = dict(a=MCMC.Post.mean.a, b=MCMC.Post.mean.b, beta=np.asarray(MCMC.Post.mean.beta).reshape(-1)[1:])
    for name, ch in chains.items():
        z = np.abs(got[name] - ch.mean(0)) / (np.sqrt(2.0) * mcse(ch))
        assert z.max() < 4.0, (name, precision, z.max())
