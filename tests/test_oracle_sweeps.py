"""Sweep-level checks of the oracle: committed golden traces (build-owned, tests/golden/make_golden.py), parameter recovery in
the style of the reference's only validation method (README.md:61-77, `test SimTools.jl`:154-178), and the numerical
character of each chain (contractive vs chaotic under common random numbers)."""
import os

import numpy as np
import pytest

import parity_util as pu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(model):
    z = np.load(os.path.join(GOLD, f"{model}.npz"))
    init = {k[5:]: z[k] for k in z.files if k.startswith("init_")}
    return z, z["Y"], (z["logT"] if "logT" in z.files else None), (z["X"] if "X" in z.files else None), init


ALL_MODELS = ["mlirt", "rtirt", "latentqr", "crossqr", "null", "cross", "latent"]


@pytest.mark.parametrize("model", ALL_MODELS)
def test_oracle_reproduces_golden_traces(model):
    z, Y, logT, X, init = load_golden(model)
    op = pu.OracleProblem(model, Y, logT, X, init, qRt=float(z["qRt"]), cov2one=(model not in ("latentqr", "latent")), seed=int(z["seed"]))
    tr = op.run(int(z["T"]), with_nu=(model == "latentqr"))
    tol = 1e-6 if model == "crossqr" else 1e-9      # CrossQr amplifies libm-level differences (see test below)
    for k in ("ra", "rt", "qr", "ll"):
        assert np.allclose(tr[k], z[k], rtol=tol, atol=tol), k


@pytest.mark.parametrize("model", ["mlirt", "rtirt"])
def test_parameter_recovery(model):
    N, J, T = 1000, 15, 300
    Y, logT, X, init, tp = pu.make_problem(model, N, J, 3, seed=3, qRt=0.5)
    op = pu.OracleProblem(model, Y, logT, X, init, qRt=0.5)
    tr = op.run(T)
    ra = tr["ra"][T // 2:].mean(0)
    a, b, th = ra[N:N + J], ra[N + J:], ra[:N]
    assert np.sqrt(np.mean((a - tp.a) ** 2)) < 0.15 and np.sqrt(np.mean((b - tp.b) ** 2)) < 0.15
    assert np.corrcoef(th, tp.theta)[0, 1] > 0.85
    assert tr["ll"][-1] > tr["ll"][0]
    if model == "rtirt":
        rt = tr["rt"][T // 2:].mean(0)
        dl = rt[N:N + J] - tp.lam      # lambda absorbs the location of zeta and the generator's truncation of logT at 0 (src/SimTools.jl:169)
        assert abs(dl.mean()) < 0.3 and np.sqrt(np.mean((dl - dl.mean()) ** 2)) < 0.05
        assert np.sqrt(np.mean((rt[N + J:] - tp.sig2t) ** 2)) < 0.05
        assert np.corrcoef(rt[:N], tp.zeta)[0, 1] > 0.98
        beta = tr["qr"][T // 2:, :8].mean(0).reshape(4, 2, order="F")
        # speed coefficients are attenuated ~10 % by the generator's truncation of logT at 0 for fast subjects
        assert np.max(np.abs(beta[1:] - np.asarray(tp.beta).reshape(3, 2))) < 0.25
        assert np.all(beta[0] == 0)                  # intercept = false zeroes beta[1,:] (src/GibbsRtIrt.pl.jl:293-295)


def test_quantile_models_recover_item_parameters():
    N, J, T = 800, 12, 240
    for model in ("latentqr", "crossqr"):
        Y, logT, X, init, tp = pu.make_problem(model, N, J, 3, seed=5, qRt=0.5)
        op = pu.OracleProblem(model, Y, logT, X, init, qRt=0.5, cov2one=(model != "latentqr"))
        tr = op.run(T)
        ra = tr["ra"][T // 2:].mean(0)
        assert np.sqrt(np.mean((ra[N:N + J] - tp.a) ** 2)) < 0.2
        assert np.sqrt(np.mean((ra[N + J:] - tp.b) ** 2)) < 0.2
        assert np.all(np.isfinite(tr["ll"]))
        if model == "crossqr":
            rho = tr["qr"][T // 2:, :J].mean(0)
            assert np.corrcoef(rho, tp.rho)[0, 1] > 0.8


def test_variants_recover_parameters():
    """GibbsRtIrtNull / Cross / Latent on setData*-style data: item parameters, and each variant's own
    structural parameter (Null: cor(theta, zeta) ~ 0 with cov2one; Cross: rho; Latent: the theta coefficient of zeta's regression)."""
    N, J, T = 800, 12, 240
    for model in ("null", "cross", "latent"):
        Y, logT, X, init, tp = pu.make_problem(model, N, J, 3, seed=5, qRt=0.5)
        op = pu.OracleProblem(model, Y, logT, X, init, qRt=0.5, cov2one=(model != "latent"))
        tr = op.run(T)
        ra = tr["ra"][T // 2:].mean(0)
        assert np.sqrt(np.mean((ra[N:N + J] - tp.a) ** 2)) < 0.2, model
        assert np.sqrt(np.mean((ra[N + J:] - tp.b) ** 2)) < 0.2, model
        assert np.corrcoef(ra[:N], tp.theta)[0, 1] > 0.8
        assert np.all(np.isfinite(tr["ll"])) and tr["ll"][-1] > tr["ll"][0]
        qr = tr["qr"][T // 2:].mean(0)
        if model == "null":
            assert np.all(tr["qr"][:, :8] == 0) and np.all(tr["qr"][:, 8] == 1) and np.all(tr["qr"][:, 11] == 1) and abs(qr[9]) < 0.2
        if model == "cross":
            assert np.corrcoef(qr[:J], tp.rho)[0, 1] > 0.8
        if model == "latent":
            bt = np.asarray(tp.beta).ravel()
            # covariate effects are recovered; theta's coefficient is attenuated (theta enters as a noisy draw, and the theta
            # conditional ignores the regression -- drawSubjAbilityNull), so only its sign is asserted
            assert qr[0] == 0 and np.max(np.abs(qr[1:4] - bt[:3])) < 0.1 and qr[4] > 0


def _perturbed_pair(model, T):
    Y, logT, X, init, _ = pu.make_problem(model, 777, 13, 3, seed=7, qRt=0.85)
    a = pu.OracleProblem(model, Y, logT, X, init, qRt=0.85, cov2one=(model not in ("latentqr", "latent")))
    init2 = dict(init)
    key = "zeta" if model != "mlirt" else "theta"
    init2[key] = np.nextafter(init[key], np.inf)          # every entry moved by exactly 1 ulp
    b = pu.OracleProblem(model, Y, logT, X, init2, qRt=0.85, cov2one=(model not in ("latentqr", "latent")))
    ta, tb = a.run(T), b.run(T)
    return [max(pu.rel_err(ta["ra"][t], tb["ra"][t]).max(), pu.rel_err(ta["rt"][t], tb["rt"][t]).max()) for t in range(T)]


@pytest.mark.parametrize("model", ["mlirt", "rtirt", "latentqr", "null", "cross", "latent"])
def test_chains_contract_under_common_random_numbers(model):
    """Two runs started 1 ulp apart with the same counter-based variates stay within 1e-9 (relative, floor 1e-6) with no growth:
    free-running elementwise parity over many sweeps is a meaningful test for these models."""
    e = _perturbed_pair(model, 12)
    assert max(e) < 1e-9 and max(e[6:]) < 1e3 * max(max(e[:6]), 1e-13)


def test_crossqr_chain_is_chaotic():
    """GibbsRtIrtCrossQr amplifies a 1-ulp difference to >1e-6 within 12 sweeps (the 1/nu weights pin zeta_i to the residual of
    its smallest-nu cell).  This is a property of the reference's algorithm; parity tests therefore check CrossQr
    free-running over 3 sweeps and teacher-forced afterwards (tests/test_gpu_parity.py)."""
    e = _perturbed_pair("crossqr", 12)
    assert e[0] < 1e-8 and max(e) > 1e-6


def test_openmp_oracle_is_bit_identical_to_the_single_thread_run():
    # the multi-threaded CPU baseline of bench.py must be the same computation: parallel loops only over independent
    # subjects / items / cells with counter-addressed draws (the log-likelihood reduction is the documented exception)
    Y, logT, X, init, _ = pu.make_problem("rtirt", 300, 7, 2, seed=5)
    lib = pu.oracle()
    try:
        lib.orc_set_threads(1)
        a = pu.OracleProblem("rtirt", Y, logT, X, init).run(4)
        lib.orc_set_threads(4)
        b = pu.OracleProblem("rtirt", Y, logT, X, init).run(4)
    finally:
        lib.orc_set_threads(1)
    for k in ("ra", "rt", "qr"):
        assert np.array_equal(a[k], b[k]), k
    assert np.allclose(a["ll"], b["ll"], rtol=1e-12)


def test_oracle_on_the_reference_demo_data():
    """data/demo.csv of the reference (its `test SimTools.jl`:186-203 fits GibbsRtIrtNull to it): the oracle's chain is finite, keeps
    Sigma_p on the unit diagonal (cov2one) and improves the log-likelihood from the constructor's initial values."""
    rows = np.genfromtxt(os.path.join(GOLD, "demo.csv"), delimiter=",", skip_header=1, usecols=range(1, 25))
    Y, logT, X = rows[:, :10].astype(np.uint8), rows[:, 10:20], rows[:, 20:24]
    g = np.random.default_rng(0)
    init = dict(theta=g.standard_normal(300), zeta=g.standard_normal(300), sigp=np.eye(2))
    tr = pu.OracleProblem("null", Y, logT, X, init, qRt=0.5, cov2one=True).run(150)
    assert np.all(np.isfinite(tr["ra"])) and np.all(np.isfinite(tr["rt"])) and np.all(np.isfinite(tr["ll"]))
    assert np.all(tr["qr"][:, :10] == 0) and np.all(tr["qr"][:, 10] == 1) and np.all(tr["qr"][:, 13] == 1)
    assert tr["ll"][-30:].mean() > tr["ll"][:3].mean()
    lam = tr["rt"][75:, 300:310].mean(0)
    assert np.max(np.abs(lam - logT.mean(0))) < 0.15          # lambda_j tracks the item's mean log time (zeta is centred near 0)
