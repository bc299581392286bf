"""Distributional cross-check of the oracle's Polya-Gamma sampler against the algorithm family the reference calls.

`drawRaPgRandomVariable` (/root/reference/src/Draw.pl.jl:38) draws omega ~ PG(1, eta) with `PolyaGammaPSWSampler(1, eta)` (PolyaGammaSamplers.jl,
absent from /root/reference): Devroye's (2009) / Polson, Scott & Windle's (2013) TWO-LEVEL sampler for J*(1, z = |eta|/2), truncation t = 0.64.
The oracle (oracle/orc_rng.h) and the device implement a single-level re-derivation with a different left-piece proposal, one hand writing both.
This file restates the classic sampler as SURVEY.md Appendix A.1 states it -- mixture weight with the inverse-Gaussian cdf, exponential-pair
rejection for z < 1/t, Michael-Schucany-Haas draws until X < t otherwise, the a_n(x) alternating series in its original (un-normalised) form --
in numpy, with numpy's own generators, sharing no code and no derivation with the oracle, and compares 10^6 draws from each by two-sample
Kolmogorov-Smirnov and Anderson-Darling tests.  It pins "same law as the reference's sampler" on something written from the literature only."""
import numpy as np
import pytest
from scipy import stats

import parity_util as pu

T = 0.64


def _a_n(n, x):
    """Coefficients of the alternating series for the Jacobi-type density J*(1, 0) (PSW 2013, eq. 14-16)."""
    k = n + 0.5
    left = np.pi * k * (2.0 / (np.pi * x)) ** 1.5 * np.exp(-2.0 * k * k / x)
    right = np.pi * k * np.exp(-0.5 * k * k * np.pi ** 2 * x)
    return np.where(x <= T, left, right)


def _ig_cdf(x, z):
    """cdf of IG(mu = 1/z, lambda = 1) at x (z > 0); the second term in logs so that e^{2z} cannot overflow."""
    r = np.sqrt(x)
    return stats.norm.cdf((x * z - 1.0) / r) + np.exp(2.0 * z + stats.norm.logcdf(-(x * z + 1.0) / r))


def _trunc_ig(z, n, g):
    """n draws of IG(1/z, 1) truncated to (0, t)  (PSW 2013, Algorithm 3 / Devroye 2009)."""
    out = np.empty(n)
    todo = np.arange(n)
    if z < 1.0 / T:
        while todo.size:
            m = todo.size
            # exponential pair: E, E' until E^2 <= 2 E'/t; X = t / (1 + t E)^2
            E = np.empty(m)
            need = np.arange(m)
            while need.size:
                e1, e2 = g.exponential(size=need.size), g.exponential(size=need.size)
                ok = e1 * e1 <= 2.0 * e2 / T
                E[need[ok]] = e1[ok]
                need = need[~ok]
            X = T / (1.0 + T * E) ** 2
            keep = g.uniform(size=m) <= np.exp(-0.5 * z * z * X)
            out[todo[keep]] = X[keep]
            todo = todo[~keep]
    else:
        mu = 1.0 / z
        while todo.size:
            m = todo.size
            y = g.standard_normal(m) ** 2
            X = mu + 0.5 * mu * mu * y - 0.5 * mu * np.sqrt(4.0 * mu * y + (mu * y) ** 2)
            flip = g.uniform(size=m) > mu / (mu + X)
            X = np.where(flip, mu * mu / X, X)
            keep = X < T
            out[todo[keep]] = X[keep]
            todo = todo[~keep]
    return out


def psw_pg1(c, n, seed):
    """n draws of PG(1, c) = J*(1, |c|/2) / 4 by the two-level PSW / Devroye sampler."""
    g = np.random.default_rng(seed)
    z = 0.5 * abs(c)
    K = np.pi ** 2 / 8.0 + 0.5 * z * z
    p = np.pi / (2.0 * K) * np.exp(-K * T)
    q = 2.0 * np.exp(-z) * _ig_cdf(T, z) if z > 0 else 4.0 * stats.norm.cdf(-1.0 / np.sqrt(T))
    out = np.empty(n)
    todo = np.arange(n)
    while todo.size:
        m = todo.size
        tail = g.uniform(size=m) < p / (p + q)
        X = np.empty(m)
        X[tail] = T + g.exponential(size=int(tail.sum())) / K
        X[~tail] = _trunc_ig(z, int((~tail).sum()), g)
        S = _a_n(0, X)
        Y = g.uniform(size=m) * S
        state = np.zeros(m, dtype=np.int8)          # 0 undecided, 1 accept, -1 reject
        nn = 0
        while np.any(state == 0) and nn < 60:
            nn += 1
            und = state == 0
            if nn & 1:
                S = np.where(und, S - _a_n(nn, X), S)
                state[und & (Y <= S)] = 1
            else:
                S = np.where(und, S + _a_n(nn, X), S)
                state[und & (Y > S)] = -1
        acc = state == 1
        out[todo[acc]] = 0.25 * X[acc]
        todo = todo[~acc]
    return out


N = 1_000_000


# c = 3.12 / 3.13 straddle z = 1/t, where the classic sampler switches its truncated-IG method; 1.99 / 2.0 and 15.99 / 16.0 straddle a bin edge of
# the oracle's proposal table and its switch to the IG proposal (z = 8)
@pytest.mark.parametrize("c", [0.0, 0.5, 1.99, 2.0, 3.12, 3.13, 6.0, 12.0, 15.99, 16.0, 24.0])
def test_oracle_pg_draws_have_the_law_of_the_classic_psw_sampler(c):
    ref = psw_pg1(c, N, seed=int(c * 100) + 11)
    got = pu.orc_sample(3, N, np.full(N, c), seed=21, sweep=int(c * 100) + 1)
    # closed-form moments of PG(1, c) for both: the classic sampler itself must be right
    m = 0.25 if c == 0 else np.tanh(c / 2) / (2 * c)
    v = 1 / 24 if c == 0 else (np.sinh(c) - c) / (4 * c ** 3 * np.cosh(c / 2) ** 2)
    for x in (ref, got):
        assert abs(x.mean() - m) < 4.5 * np.sqrt(v / N)
    assert stats.ks_2samp(ref, got).pvalue > 1e-3
    ad = stats.anderson_ksamp([ref[:200_000], got[:200_000]])
    assert ad.statistic < 6.546, ad             # the 0.1 % critical value (scipy caps the reported significance level there)


def test_classic_sampler_detects_a_wrong_law():
    """Power check of the comparison itself: the same test rejects draws from PG(1, c') with c' 3 % off."""
    ref = psw_pg1(2.0, N, seed=3)
    off = pu.orc_sample(3, N, np.full(N, 2.06), seed=21, sweep=5)
    assert stats.ks_2samp(ref, off).pvalue < 1e-3
