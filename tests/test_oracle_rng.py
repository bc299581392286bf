"""Build-owned pins of the oracle's random streams and scalar samplers ("parity unpinned": the reference holds no golden
vectors for this path, so every sampler is checked against closed-form answers)."""
import ctypes as C

import numpy as np
import pytest
from scipy import special, stats

import parity_util as pu

N = 400_000


def _philox(ctr, key):
    c, k, o = np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
    pu.oracle().orc_philox(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return [int(x) for x in o]


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert _philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_streams_are_addressed_by_site_index_sweep():
    a = pu.orc_sample(0, 1000, seed=1, site=3, sweep=7)
    assert np.array_equal(a, pu.orc_sample(0, 1000, seed=1, site=3, sweep=7))
    for kw in (dict(seed=2, site=3, sweep=7), dict(seed=1, site=4, sweep=7), dict(seed=1, site=3, sweep=8)):
        assert not np.array_equal(a, pu.orc_sample(0, 1000, **kw))
    assert len(np.unique(a)) == 1000


def test_uniform_normal_expo_distributions():
    u, z, e = pu.orc_sample(0, N), pu.orc_sample(1, N), pu.orc_sample(2, N)
    assert 0 < u.min() and u.max() < 1
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    assert stats.kstest(z, "norm").pvalue > 1e-3
    assert stats.kstest(e, "expon").pvalue > 1e-3


def test_ndtri_matches_scipy():
    p = np.concatenate([np.logspace(-11, -0.31, 3000), 1 - np.logspace(-9, -0.31, 3000)])
    got = pu.orc_sample(9, p.size, p)
    assert np.max(np.abs(got - special.ndtri(p)) / np.abs(special.ndtri(p))) < 1e-9


@pytest.mark.parametrize("c", [0.0, 0.3, 1.0, 2.0, 3.1, 3.13, 4.0, 7.0, 15.0, 40.0])
def test_polya_gamma_moments(c):
    """PG(1,c): E = tanh(c/2)/(2c), Var = (sinh c - c)/(4 c^3 cosh^2(c/2))."""
    x = pu.orc_sample(3, N, np.full(N, c), seed=11, sweep=int(c * 10) + 1)
    m = 0.25 if c == 0 else np.tanh(c / 2) / (2 * c)
    v = 1 / 24 if c == 0 else ((np.sinh(c) - c) / (4 * c ** 3 * np.cosh(c / 2) ** 2) if c < 30 else 1 / (2 * c ** 3))
    assert abs(x.mean() - m) < 4.5 * np.sqrt(v / N)
    m4 = np.mean((x - m) ** 4)
    assert abs(x.var() - v) < 4.5 * np.sqrt((m4 - v * v) / N)
    assert x.min() > 0


def test_polya_gamma_laplace_transform():
    """E exp(-t w) for w ~ PG(1, c) is cosh(c/2) / cosh(sqrt((c^2/2 + t)/2)) -- checks the whole law, not two moments."""
    for c in (0.0, 1.5, 3.5):
        x = pu.orc_sample(3, N, np.full(N, c), seed=5, sweep=3)
        for t in (0.5, 2.0, 8.0):
            want = np.cosh(c / 2) / np.cosh(np.sqrt((c * c / 2 + t) / 2))
            got = np.exp(-t * x)
            assert abs(got.mean() - want) < 4.5 * got.std() / np.sqrt(N)


def test_inverse_gaussian_and_qr_weight():
    """IG(mu, lambda): mean mu, var mu^3/lambda, cdf via scipy (invgauss(mu/lambda, scale=lambda)).  1/IG(mu, lambda) is the
    GIG(p=+1/2) variate the dead-code GenInvGaussian sampler (src/GenInvGaussian.jl:76-106) would produce."""
    for mu, lam in ((0.5, 1.0), (2.0, 3.0), (30.0, 0.7)):
        x = pu.orc_sample(4, N, np.full(N, mu), np.full(N, lam), seed=3, sweep=int(mu * 7) + 1)
        assert abs(x.mean() - mu) < 4.5 * np.sqrt(mu ** 3 / lam / N)
        assert stats.kstest(x, stats.invgauss(mu / lam, scale=lam).cdf).pvalue > 1e-3
    pa, pb = 0.8, 1.7
    nu = pu.orc_sample(8, N, np.full(N, pa), np.full(N, pb))
    assert np.all((nu >= 1e-10) & (nu <= 1e10))
    assert stats.kstest(1 / nu, stats.invgauss((pb / pa) / pb ** 2, scale=pb ** 2).cdf).pvalue > 1e-3
    assert np.all(np.isfinite(pu.orc_sample(8, 1000, np.full(1000, 1e-300), np.full(1000, 1.7))))   # mu clamp path


@pytest.mark.parametrize("m,s", [(1.0, 0.2), (0.0, 1.0), (-2.0, 0.5), (-8.0, 1.0), (4.0, 0.01)])
def test_truncated_normal(m, s):
    x = pu.orc_sample(5, N, np.full(N, m), np.full(N, s), seed=9)
    assert x.min() > 0
    assert stats.kstest(x, stats.truncnorm((0 - m) / s, np.inf, loc=m, scale=s).cdf).pvalue > 1e-3


@pytest.mark.parametrize("shape", [1.0, 2.5, 50.0, 5e4, 7.5e5])
def test_gamma(shape):
    x = pu.orc_sample(6, 200_000, np.full(200_000, shape), seed=4)
    assert stats.kstest(x, stats.gamma(shape).cdf).pvalue > 1e-3


@pytest.mark.parametrize("p,a,b", [(0.5, 2.0, 3.0), (-0.5, 1.0, 1.0), (2.5, 0.7, 4.0), (-3.0, 5.0, 0.2), (10.0, 3.0, 1.0)])
def test_gig_moments_match_bessel_ratios(p, a, b):
    """GIG(p, a, b) of src/GenInvGaussian.jl (density ~ x^(p-1) exp(-(a x + b/x)/2)): E[X^k] = (b/a)^(k/2) K_{p+k}(w) / K_p(w), w = sqrt(ab)
    (the reference's own mean/var formulas, GenInvGaussian.jl:36-52); p = -1/2 is the inverse Gaussian of the live quantile weights."""
    import ctypes as C
    from scipy.special import kv
    lib = pu.oracle()
    lib.orc_sample_gig.argtypes = [C.c_uint64, C.c_int, C.c_uint32, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_void_p]
    lib.orc_sample_gig.restype = None
    n = 200_000
    x = np.empty(n)
    lib.orc_sample_gig(1234, 15, 1, n, p, a, b, x.ctypes.data)
    w = np.sqrt(a * b)
    for k in (1, 2, -1):
        want = (b / a) ** (k / 2) * kv(p + k, w) / kv(p, w)
        sd = np.std(x ** k) / np.sqrt(n)
        assert abs(np.mean(x ** k) - want) < 5 * sd, (k, np.mean(x ** k), want)
    if p == -0.5:                              # GIG(-1/2, a, b) == InverseGaussian(mu = sqrt(b/a), lambda = b)
        ig = pu.orc_sample(4, n, np.full(n, np.sqrt(b / a)), np.full(n, b), seed=99)
        from scipy.stats import ks_2samp
        assert ks_2samp(x, ig).pvalue > 1e-3
