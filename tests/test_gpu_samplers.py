"""Device samplers vs the oracle's: same (seed, site, i, sweep) stream => same variate.
fp64: agreement to rounding (1e-12 relative); fp32: the fp32 evaluation of the same algorithm (tolerances stated per test)."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu

N = 200_000


def _dev(which, n, p0=None, p1=None, precision=1, **kw):
    L = pu.ge.load_package()._lib
    return L.debug_sample(which, n, p0, p1, precision=precision, **kw)


@pytest.mark.parametrize("which,name", [(0, "uniform"), (1, "normal"), (2, "expo")])
def test_basic_variates_f64(which, name):
    d, o = _dev(which, N), pu.orc_sample(which, N)
    assert np.max(np.abs(d - o)) < 1e-12, name


def test_uniform_f32_is_rounded_f64():
    d, o = _dev(0, N, precision=0), pu.orc_sample(0, N)
    assert np.max(np.abs(d - o)) <= 2.0 ** -23
    assert d.min() > 0.0 and d.max() < 1.0


def test_pg_mixture_weight():
    z = np.linspace(0.0, 12.0, 4001)
    o = pu.orc_sample(7, z.size, z)
    assert np.max(np.abs(_dev(7, z.size, z) - o)) < 1e-12
    assert np.max(np.abs(_dev(7, z.size, z, precision=0) - o)) < 2e-5     # fp32 evaluation of exp(x0 -+ z + log Phi)


def test_pg1_f64_matches_oracle():
    g = np.random.default_rng(0)
    c = g.normal(0, 2.5, N)
    d, o = _dev(3, N, c), pu.orc_sample(3, N, c)
    assert np.max(np.abs(d - o) / o) < 1e-10


def test_pg1_f64_filtered_decisions_are_the_reference_decisions():
    """The fp64 engine's attempt takes its accept / reject decisions in fp32 behind guard bands and evaluates only the accepted value in
    fp64 (erm_rng.hpp, pg1_filter + pg1_value_*).  Against the reference form of the attempt (every statement in fp64) on 2^26 draws:
    a flipped decision would show as a different draw altogether; rounding of the value alone stays below 1e-13."""
    n = 1 << 24
    g = np.random.default_rng(5)
    for k, c in enumerate((g.normal(0, 2.5, n), g.uniform(-3.2, 3.2, n), g.uniform(3.0, 7.0, n), np.concatenate([g.uniform(19.0, 26.0, n // 2), g.uniform(0, 1e-3, n // 2)]))):
        fast, ref = _dev(3, n, c, sweep=11 + k), _dev(10, n, c, sweep=11 + k)
        assert np.max(np.abs(fast - ref) / ref) < 1e-13
    o = pu.orc_sample(3, N, c[:N], sweep=14)
    assert np.max(np.abs(fast[:N] - o) / o) < 1e-10


def test_ndtri_f64_is_as241_and_matches_the_oracles_newton_form():
    from scipy.special import ndtri
    g = np.random.default_rng(6)
    p = np.concatenate([g.uniform(0, 1, N), 10.0 ** g.uniform(-11, -1, N), 1.0 - 10.0 ** g.uniform(-11, -1, N), np.linspace(0.07, 0.08, 1001), [0.075, 0.925, 0.5]])
    d = _dev(9, p.size, p)
    ref = ndtri(p)
    assert np.max(np.abs(d - ref) / np.maximum(np.abs(ref), 0.1)) < 1e-14        # scipy (Cephes): independent of both implementations
    o = pu.orc_sample(9, p.size, p)
    assert np.max(np.abs(d - o) / np.maximum(np.abs(o), 0.1)) < 1e-13


def test_fp64_cell_path_elementary_functions():
    """fm::log / exp_neg / sqrt / div (erm_rng.hpp): range-specialised fp64 forms of the cell path, against numpy (libm)."""
    g = np.random.default_rng(8)
    x = np.concatenate([g.uniform(0, 1, N), 10.0 ** g.uniform(-12, 2, N), 1.0 + g.uniform(-1e-6, 1e-6, 1000), [1.0, 0.5, 2.0, 2.0 ** -0.5]])
    d = _dev(11, x.size, x)
    assert np.max(np.abs(d - np.log(x)) / np.maximum(np.abs(np.log(x)), 1e-8)) < 4e-16 * 8
    assert _dev(11, 1, np.array([1.0]))[0] == 0.0
    t = _dev(15, x.size, x)                                     # the LDS-table form: absolute accuracy (its callers add to / take roots of O(1) values)
    assert np.max(np.abs(t - np.log(x)) / np.maximum(np.abs(np.log(x)), 1.0)) < 5e-16
    a = np.concatenate([g.uniform(0, 40, N), g.uniform(0, 1e-3, 1000), g.uniform(600, 800, 1000), [0.0]])
    d = _dev(12, a.size, a)
    ref = np.exp(-np.minimum(a, 700.0))
    assert np.max(np.abs(d - ref) / ref) < 2e-15
    d = _dev(18, a.size, a)                                     # the cell log-likelihood's shorter polynomial: absolute accuracy on a value that is added to 1
    assert np.max(np.abs(d - ref)) < 3e-16 and np.max(np.abs(d - ref) / ref) < 4e-14
    w = np.concatenate([g.integers(0, 2 ** 32, N), np.arange(0, 4096), 2 ** 32 - 1 - np.arange(0, 4096), 2 ** np.arange(0, 32)]).astype(np.float64)
    d = _dev(19, w.size, w)                                     # -log u1 of a PG attempt straight from its word: the same bits as the table logarithm of the uniform
    assert np.array_equal(d, _dev(15, w.size, (w + 0.5) * 2.0 ** -32))
    d = _dev(13, x.size, x)
    assert np.max(np.abs(d - np.sqrt(x)) / np.sqrt(x)) < 3e-16
    u = np.concatenate([g.uniform(0, 1, N), (np.arange(0, 4097) / 4096.0), (np.arange(0, 2 ** 12) + 0.5) * 2.0 ** -32, 1 - (np.arange(0, 2 ** 12) + 0.5) * 2.0 ** -32])
    d = _dev(16, u.size, u)
    ref = np.cos(2 * np.longdouble("3.14159265358979323846264338327950288") * u.astype(np.longdouble)).astype(np.float64)      # 80-bit argument: 2 pi u in double is already off by 7e-16
    assert np.max(np.abs(d - ref)) < 3e-16
    y = 10.0 ** g.uniform(-6, 6, x.size) * g.choice([-1.0, 1.0], x.size)
    d = _dev(14, x.size, y, x)
    assert np.max(np.abs(d - y / x) / np.abs(y / x)) < 3e-16


def test_pg1_f32_matches_oracle_except_rare_flips():
    g = np.random.default_rng(1)
    c = g.normal(0, 2.5, N)
    d, o = _dev(3, N, c, precision=0), pu.orc_sample(3, N, c)
    rel = np.abs(d - o) / o
    assert np.mean(rel > 1e-4) < 2e-4          # accept/reject decisions flip only where fp32 rounding crosses a threshold
    assert np.median(rel) < 1e-6
    # and the fp32 draws still have the right moments
    for cc in (0.0, 1.0, 4.0):
        x = _dev(3, N, np.full(N, cc), precision=0, sweep=3)
        m = 0.25 if cc == 0 else np.tanh(cc / 2) / (2 * cc)
        v = 1 / 24 if cc == 0 else (np.sinh(cc) - cc) / (4 * cc ** 3 * np.cosh(cc / 2) ** 2)
        assert abs(x.mean() - m) < 5 * np.sqrt(v / N)


def test_invgauss_and_qr_weight():
    g = np.random.default_rng(2)
    mu, lam = np.exp(g.normal(0, 1.5, N)), np.exp(g.normal(0, 1, N))
    d, o = _dev(4, N, mu, lam), pu.orc_sample(4, N, mu, lam)
    assert np.max(np.abs(d - o) / o) < 1e-10
    d32 = _dev(4, N, mu, lam, precision=0)
    assert np.mean(np.abs(d32 - o) / o > 1e-3) < 1e-3
    pa, pb = np.abs(g.normal(0, 1, N)) + 1e-12, np.full(N, 1.7)
    d, o = _dev(8, N, pa, pb), pu.orc_sample(8, N, pa, pb)
    assert np.max(np.abs(d - o) / o) < 1e-10


def test_qr_weight_cell_path_form_is_the_same_variate():
    """The fp64 engine's cell path draws the quantile weight through range-specialised division / root / reciprocal (qr_weight_q: debug sampler 17) with
    parB / parA = cB / |residual| and lambda = parB^2 hoisted per item; same stream, same law, same variate as the generic form (8) and as the oracle,
    including exactly zero and denormal-sized residuals (infinite mean -> the Levy limit) and both clamps."""
    g = np.random.default_rng(12)
    pa = np.concatenate([np.abs(g.normal(0, 1, N)) + 1e-12, 10.0 ** g.uniform(-12, 3, N), np.zeros(64), np.full(64, 1e-30), np.full(64, 1e-200), np.full(64, 1e8)])
    pb = np.concatenate([np.full(N, 1.7), np.exp(g.normal(0.5, 0.7, N)), np.full(256, 1.7)])
    d, o = _dev(17, pa.size, pa, pb), pu.orc_sample(8, pa.size, pa, pb)
    assert np.all(np.isfinite(d)) and np.all((d >= 1e-10) & (d <= 1e10))
    assert np.max(np.abs(d - o) / o) < 1e-9
    assert np.max(np.abs(_dev(8, pa.size, pa, pb) - o) / o) < 1e-9


def test_item_level_samplers():
    g = np.random.default_rng(3)
    m, s = g.normal(0.5, 2, N), np.exp(g.normal(-1, 1, N))
    d, o = _dev(5, N, m, s), pu.orc_sample(5, N, m, s)
    assert np.max(np.abs(d - o) / (np.abs(m) + s)) < 1e-12      # m + s*z cancels near the truncation point
    assert d.min() > 0
    sh = np.exp(g.uniform(0, 13, 20000))        # shapes 1 .. 4e5 (N/2 and 3N/2 at the benchmark sizes)
    d, o = _dev(6, sh.size, sh), pu.orc_sample(6, sh.size, sh)
    assert np.max(np.abs(d - o) / o) < 1e-10


def test_qr_weight_survives_degenerate_residuals():
    """parA == 0 (an exactly zero residual: measure-zero in fp64, reachable in fp32) gives mu = inf; the IG root must stay
    finite (Levy limit) instead of NaN -- the reference's expression src/Draw.pl.jl:310-312 would propagate a NaN here."""
    n = 4096
    pa = np.zeros(n)
    pa[::2] = 1e-30
    pb = np.full(n, 1.7)
    for prec in (0, 1):
        d = _dev(8, n, pa, pb, precision=prec)
        assert np.all(np.isfinite(d)) and np.all((d >= 1e-10) & (d <= 1e10))
    o = pu.orc_sample(8, n, pa, pb)
    assert np.all(np.isfinite(o))
    assert np.max(np.abs(_dev(8, n, pa, pb) - o) / o) < 1e-9


@pytest.mark.parametrize("p,a,b", [(0.5, 2.0, 3.0), (-0.5, 1.0, 1.0), (2.5, 0.7, 4.0), (-3.0, 5.0, 0.2), (0.1, 0.01, 0.02)])
def test_gig_matches_oracle(p, a, b):
    """erm_sample_gig (general-p GIG of src/GenInvGaussian.jl, Devroye 2014) against the oracle's restatement, draw by draw."""
    import ctypes as C
    lib = pu.oracle()
    lib.orc_sample_gig.argtypes = [C.c_uint64, C.c_int, C.c_uint32, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_void_p]
    lib.orc_sample_gig.restype = None
    n = 20000
    want = np.empty(n)
    lib.orc_sample_gig(1234, 15, 1, n, p, a, b, want.ctypes.data)
    got = pu.ge.load_package()._lib.sample_gig(p, a, b, n, seed=1234, site=15, sweep=1)
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-9
