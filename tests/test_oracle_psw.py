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

from psw_classic import psw_pg1


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
