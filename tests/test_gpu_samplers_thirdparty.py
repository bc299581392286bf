"""The DEVICE samplers against code the builder did not write: scipy.stats distributions and the classic two-level Polya-Gamma sampler of the
literature (tests/psw_classic.py, numpy generators), >= 10^6 device draws per case, Kolmogorov-Smirnov / Anderson-Darling.

The oracle and the kernels share one builder-owned sampling specification, so agreement between them -- however exact --
cannot catch a wrong LAW; the reference's own samplers live in packages that are absent here (PolyaGammaSamplers 0.1, Distributions 0.25).  These
tests pin the law of every device sampler on the GPU box, where the driver runs them.  Call sites pinned (/root/reference/src/Draw.pl.jl):
  :38   PolyaGammaPSWSampler(1, eta)            -> debug sampler 3 (fp64 engine and fp32 fast mode), 10 (the all-fp64 reference form of the attempt)
  :91   truncated(Normal(m, s), 0, Inf)         -> 5            :218 (lambda: the same truncated normal)
  :260  InverseGamma(shape, scale)              -> scale / 6    (Marsaglia-Tsang gamma; shapes up to 3N/2)
  :312, :335  1 / InverseGaussian(mu, lambda)   -> 4, 8, 17     (Michael-Schucany-Haas; the quantile weights of CrossQr / LatentQr)
  :505  InverseWishart(N + 3, Psi), 2 x 2       -> erm_debug_invwishart (the structural step's own device functions)
  src/GenInvGaussian.jl:17-30 GIG(p, a, b)      -> erm_sample_gig
A fixed seed makes every test deterministic: a p-value threshold of 1e-4 is a statement about these draws, not a flaky gate."""
import numpy as np
import pytest
from scipy import stats

import parity_util as pu
from psw_classic import psw_pg1

pytestmark = pytest.mark.gpu

N = 1_000_000
PMIN = 1e-4


def _L():
    return pu.ge.load_package()._lib


def _dev(which, n, p0=None, p1=None, precision=1, **kw):
    return _L().debug_sample(which, n, None if p0 is None else np.full(n, float(p0)), None if p1 is None else np.full(n, float(p1)), precision=precision, **kw)


import functools


@functools.lru_cache(maxsize=None)
def _psw(c, seed):
    return psw_pg1(c, N, seed=seed)


def _pg_moments(c):
    m = 0.25 if c == 0 else np.tanh(c / 2) / (2 * c)
    v = 1 / 24 if c == 0 else (np.sinh(c) - c) / (4 * c ** 3 * np.cosh(c / 2) ** 2)
    return m, v


# c = 3.13 straddles z = 1/t (the classic sampler's switch of truncated-IG method), 2.0 a bin edge of the device's proposal table, 16+ the device's
# switch to the IG proposal (z >= 8)
@pytest.mark.parametrize("precision", [1, 0], ids=["f64", "f32"])
@pytest.mark.parametrize("c", [0.0, 0.5, 2.0, 3.13, 6.0, 12.0, 24.0])
def test_device_pg_draws_have_the_law_of_the_classic_psw_sampler(c, precision):
    ref = _psw(c, int(c * 100) + 911)
    got = _dev(3, N, c, precision=precision, seed=77, sweep=int(c * 100) + 3)
    m, v = _pg_moments(c)
    for x in (ref, got):
        assert abs(x.mean() - m) < 4.5 * np.sqrt(v / N)
    assert abs(got.var() / v - 1.0) < 0.02
    assert stats.ks_2samp(ref, got).pvalue > PMIN
    ad = stats.anderson_ksamp([ref[:200_000], got[:200_000]])
    assert ad.statistic < 6.546, ad             # the 0.1 % critical value (scipy caps the reported significance level there)


def test_device_pg_reference_form_and_the_power_of_the_comparison():
    """The all-fp64 form of the attempt (what the guard-band cells of the fp64 engine fall back to) has the same law; and the comparison itself
    rejects device draws of PG(1, c') with c' 3 % off."""
    ref = psw_pg1(2.0, N, seed=5)
    assert stats.ks_2samp(ref, _dev(10, N, 2.0, seed=78, sweep=9)).pvalue > PMIN
    assert stats.ks_2samp(ref, _dev(3, N, 2.06, seed=78, sweep=10)).pvalue < 1e-3


@pytest.mark.parametrize("precision", [1, 0], ids=["f64", "f32"])
@pytest.mark.parametrize("mu,lam", [(1.0, 1.0), (0.3, 2.5), (4.0, 0.7), (25.0, 3.0)])
def test_inverse_gaussian_against_scipy(mu, lam, precision):
    x = _dev(4, N, mu, lam, precision=precision, seed=31, sweep=2)
    d = stats.invgauss(mu / lam, scale=lam)         # scipy's invgauss(mu') scaled by lambda is IG(mu' lambda, lambda)
    assert stats.kstest(x, d.cdf).pvalue > PMIN
    assert abs(x.mean() / mu - 1.0) < 5 * np.sqrt(mu / lam / N) + (2e-6 if precision == 0 else 0)      # sd / mean = sqrt(mu / lambda)


@pytest.mark.parametrize("which", [8, 17])
@pytest.mark.parametrize("pa,pb", [(0.8, 1.7), (0.05, 2.4), (3.0, 0.9)])
def test_quantile_weight_is_a_reciprocal_inverse_gaussian(pa, pb, which):
    """nu = clamp(1 / IG(parB / parA, parB^2), 1e-10, 1e10) (src/Draw.pl.jl:310-318, 333-341): 1 / nu against scipy's inverse Gaussian; 8 = the generic form,
    17 = the fp64 cell path's form (range-specialised division / root)."""
    nu = _dev(which, N, pa, pb, seed=33, sweep=4)
    assert np.all((nu >= 1e-10) & (nu <= 1e10))
    mu, lam = pb / pa, pb * pb
    assert stats.kstest(1.0 / nu, stats.invgauss(mu / lam, scale=lam).cdf).pvalue > PMIN


@pytest.mark.parametrize("m,s", [(1.0, 0.2), (0.05, 0.3), (-0.4, 0.25), (-2.0, 0.5), (3.0, 2.0)])
def test_truncated_normal_against_scipy(m, s):
    """truncated(Normal(m, s), 0, Inf): the N(0,1)-rejection branch (alpha <= 0) and Robert's exponential branch (alpha > 0)."""
    x = _dev(5, N, m, s, seed=35, sweep=6)
    assert x.min() > 0
    assert stats.kstest(x, stats.truncnorm((0.0 - m) / s, np.inf, loc=m, scale=s).cdf).pvalue > PMIN


@pytest.mark.parametrize("shape", [1.0, 2.5, 50.0, 500.5, 50_001.5, 750_000.001])
def test_gamma_and_inverse_gamma_against_scipy(shape):
    """Marsaglia-Tsang Gamma(shape, 1) and InverseGamma(shape, scale) = scale / Gamma (src/Draw.pl.jl:260: shapes N/2 and 3N/2 up to 750 000)."""
    g = _dev(6, N, shape, seed=37, sweep=8)
    assert stats.kstest(g, stats.gamma(shape).cdf).pvalue > PMIN
    scale = 0.3 * shape
    assert stats.kstest(scale / g, stats.invgamma(shape, scale=scale).cdf).pvalue > PMIN


def _gig_cdf(p, a, b):
    """cdf of GIG(p, a, b) ~ x^(p-1) exp(-(a x + b / x) / 2) (src/GenInvGaussian.jl:17-30) by Simpson's rule on the density of log X over a dense grid, normalised
    by the Bessel function of its closed form.  (scipy.stats.geninvgauss integrates per point with quad and returns isolated wrong -- non-monotone -- values for
    small parameters such as (0.1, 0.01, 0.02): its cdf is cross-checked below at a few quantiles by the median of neighbouring evaluations, not used for the KS test.)"""
    from scipy import integrate, special
    om = np.sqrt(a * b)
    lognorm = 0.5 * p * np.log(a / b) - np.log(2.0 * special.kve(p, om)) + om           # kve = kv * e^om
    u = np.linspace(-80.0, 40.0, 1_200_001)
    f = np.exp(lognorm + p * u - 0.5 * (a * np.exp(u) + b * np.exp(-u)))               # density of U = log X
    F = integrate.cumulative_simpson(f, x=u, initial=0.0)
    assert abs(F[-1] - 1.0) < 1e-9
    return lambda x: np.interp(np.log(x), u, F)


@pytest.mark.parametrize("p,a,b", [(0.5, 2.0, 3.0), (-0.5, 1.0, 1.0), (2.5, 0.7, 4.0), (-3.0, 5.0, 0.2), (0.1, 0.01, 0.02)])
def test_gig_against_its_density_and_scipy(p, a, b):
    """GIG(p, a, b) = scipy's geninvgauss(p, sqrt(a b)) scaled by sqrt(b / a): KS against the integrated density, the mean against the Bessel ratio."""
    from scipy import special
    x = _L().sample_gig(p, a, b, N, seed=39, sweep=10)
    cdf = _gig_cdf(p, a, b)
    assert stats.kstest(x, cdf).pvalue > PMIN
    d = stats.geninvgauss(p, np.sqrt(a * b), scale=np.sqrt(b / a))
    for q in np.quantile(x, [0.05, 0.25, 0.5, 0.75, 0.95]):
        assert abs(np.median(d.cdf(q * np.array([0.999, 1.0, 1.001]))) - cdf(q)) < 2e-4
    om = np.sqrt(a * b)
    m1 = np.sqrt(b / a) * special.kve(p + 1, om) / special.kve(p, om)
    m2 = (b / a) * special.kve(p + 2, om) / special.kve(p, om)
    assert abs(x.mean() - m1) < 5 * np.sqrt((m2 - m1 * m1) / N)


@pytest.mark.parametrize("nu,psi", [(1003.0, [[1600.0, 300.0], [300.0, 900.0]]), (13.0, [[2.0, -0.7], [-0.7, 1.5]]), (100_003.0, [[1.0e5, 2.0e4], [2.0e4, 3.0e4]])])
def test_inverse_wishart_against_scipy(nu, psi):
    """The structural step's 2 x 2 InverseWishart(N + 3, e'e + I) (src/Draw.pl.jl:499-515) through its own device functions: mean Psi / (nu - 3), the
    marginal of a diagonal entry (InverseGamma((nu - 1) / 2, Psi_kk / 2)), and every entry, the determinant and the correlation against 2 * 10^5 draws of
    scipy.stats.invwishart by two-sample KS."""
    psi = np.asarray(psi)
    n = 400_000
    S = _L().debug_invwishart(nu, psi, n, seed=41, sweep=12)
    assert np.allclose(S[:, 0, 1], S[:, 1, 0])
    mean = S.mean(axis=0)
    se = S.std(axis=0) / np.sqrt(n)
    assert np.all(np.abs(mean - psi / (nu - 3.0)) < 5 * se + 1e-12)
    for k in (0, 1):
        assert stats.kstest(S[:, k, k], stats.invgamma((nu - 1.0) / 2.0, scale=psi[k, k] / 2.0).cdf).pvalue > PMIN
    R = stats.invwishart(df=nu, scale=psi).rvs(size=200_000, random_state=np.random.default_rng(43))
    feats = lambda A: (A[:, 0, 0], A[:, 1, 1], A[:, 0, 1], A[:, 0, 0] * A[:, 1, 1] - A[:, 0, 1] ** 2, A[:, 0, 1] / np.sqrt(A[:, 0, 0] * A[:, 1, 1]))
    for x, y in zip(feats(S), feats(R)):
        assert stats.ks_2samp(x, y).pvalue > PMIN


def test_normal_uniform_exponential_against_scipy():
    for prec in (1, 0):
        assert stats.kstest(_dev(1, N, precision=prec, seed=45, sweep=14), stats.norm.cdf).pvalue > PMIN
        assert stats.kstest(_dev(2, N, precision=prec, seed=45, sweep=15), stats.expon.cdf).pvalue > PMIN
        assert stats.kstest(_dev(0, N, precision=prec, seed=45, sweep=16), stats.uniform.cdf).pvalue > PMIN
