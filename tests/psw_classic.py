"""The classic two-level Polya-Gamma sampler of the literature, in numpy -- test infrastructure shared by the CPU pin of the oracle
(tests/test_oracle_psw.py) and the GPU pin of the device samplers (tests/test_gpu_samplers_thirdparty.py).

`drawRaPgRandomVariable` (/root/reference/src/Draw.pl.jl:38) draws omega ~ PG(1, eta) with `PolyaGammaPSWSampler(1, eta)` (PolyaGammaSamplers.jl,
absent from /root/reference): Devroye's (2009) / Polson, Scott & Windle's (2013) TWO-LEVEL sampler for J*(1, z = |eta|/2), truncation t = 0.64.
This restates it as SURVEY.md Appendix A.1 states it -- mixture weight with the inverse-Gaussian cdf, exponential-pair rejection for z < 1/t,
Michael-Schucany-Haas draws until X < t otherwise, the a_n(x) alternating series in its original (un-normalised) form -- with numpy's own
generators, sharing no code and no derivation with the oracle or the device kernels."""
import numpy as np
from scipy import stats

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
