"""Sweep-level parity: the HIP path (fused schedule, sufficient statistics) against the CPU oracle (reference's un-fused
schedule, direct sums) on the same seeded inputs, through the C ABI.

Tolerances (north_star: "draws matching the reference CPU sampler under a fixed RNG seed within a stated floating-point
tolerance"):
  * fp64 engine: every traced quantity of every sweep within 1e-8 relative (floor 1e-6 absolute) -- summation order
    and expanded squares are the only differences;
  * fp32 engine: after ONE sweep >= 99.9 % of the subject draws within 2e-4, item draws within 2e-4; over many sweeps the
    common-random-number coupling keeps chains together: posterior means of item parameters within 5e-3.
"""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu

MODELS = ["mlirt", "rtirt", "latentqr", "crossqr", "null", "cross", "latent"]     # the last three: SURVEY.md 8(f).1 variants
NO_INTERCEPT = ("crossqr", "cross", "null")    # their sample! methods have no `intercept` keyword


@pytest.mark.parametrize("model", MODELS)
def test_f64_traces_match_oracle(model):
    # GibbsRtIrtCrossQr is numerically chaotic (tests/test_oracle_sweeps.py::test_crossqr_chain_is_chaotic: the oracle run
    # twice from states 1 ulp apart separates to 1e-2 within 12 sweeps), so free-running parity is checked over 3 sweeps there
    # and every later sweep is checked teacher-forced (test_f64_teacher_forced).
    res = pu.run_pair(model, N=777, J=13, nsweeps=3 if model == "crossqr" else 12, precision="f64")
    err = pu.max_rel_err(res)          # CrossQr's qr row includes vec(nu): every per-cell weight of every sweep is compared
    assert err < 1e-8, err


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("kw", [dict(intercept=True), dict(onepl=True), dict(cov2one=False), dict(cov2one=True)])
def test_f64_kwargs(model, kw):
    if model in NO_INTERCEPT and "intercept" in kw:
        pytest.skip("this model's sample! has no intercept kwarg")
    res = pu.run_pair(model, N=300, J=9, nsweeps=3 if model == "crossqr" else 6, precision="f64", **kw)
    assert pu.max_rel_err(res) < 1e-8


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("W", [1, 4, 16, 64])
def test_f64_geometry_invariance(model, W):
    """Draws are addressed by (site, i, j, sweep): lanes-per-subject, block and grid size must not change them."""
    res = pu.run_pair(model, N=401, J=21, nsweeps=3 if model == "crossqr" else 5, precision="f64", lanes_per_row=W,
                      block_threads=256, grid_blocks=3)
    assert pu.max_rel_err(res) < 1e-8


@pytest.mark.parametrize("F", [0, 3, 6])
def test_latentqr_intended_sigp_scale(F):
    """sigp_mode = 1: the evidently intended sum r_i^2/(2 k2 nu_i) of drawSubjCovarianceLatentQr (src/Draw.pl.jl:594 evaluates an
    N x N matrix division instead -- mode 0); on the device it needs the 1/nu-weighted Gram statistics of [1 X theta | u]."""
    res = pu.run_pair("latentqr", N=500, J=9, nsweeps=8, F=F, precision="f64", sigp_mode=1)
    assert pu.max_rel_err(res) < 1e-8
    ref = pu.run_pair("latentqr", N=500, J=9, nsweeps=8, F=F, precision="f64", sigp_mode=0)
    assert not np.allclose(res["dev_qr"][:, F + 2 + 3], ref["dev_qr"][:, F + 2 + 3])        # Sigp[2,2] differs between the modes


@pytest.mark.parametrize("model", ["mlirt", "rtirt", "latent", "cross"])
def test_f64_single_wave_workgroups(model):
    """block_threads = 64: one wave does everything (item draws, structural chain, Sigma_p variates, rows)."""
    res = pu.run_pair(model, N=333, J=9, nsweeps=4, precision="f64", block_threads=64, grid_blocks=7)
    assert pu.max_rel_err(res) < 1e-8


@pytest.mark.parametrize("model", ["rtirt", "latentqr", "crossqr"])
def test_f64_more_workgroups_than_compute_units(model):
    """A grid larger than the chip: late workgroups start after early ones have finished and published this sweep's parameter
    block, statistics and counters -- the fused sweep kernel's inputs are double-buffered, so they must still read last sweep's."""
    res = pu.run_pair(model, N=6000, J=7, nsweeps=3 if model == "crossqr" else 6, precision="f64", block_threads=128, grid_blocks=1500)
    assert pu.max_rel_err(res) < 1e-8


@pytest.mark.parametrize("J,N", [(1, 50), (64, 130), (65, 70), (130, 40)])
def test_f64_ragged_shapes(J, N):
    res = pu.run_pair("rtirt", N=N, J=J, nsweeps=4, precision="f64")
    assert pu.max_rel_err(res) < 1e-8


def test_nfeat_zero_and_many():
    for F in (0, 8):
        res = pu.run_pair("rtirt", N=200, J=7, nsweeps=4, F=F, precision="f64")
        assert pu.max_rel_err(res) < 1e-8
    res = pu.run_pair("mlirt", N=200, J=7, nsweeps=4, F=1, precision="f64")
    assert pu.max_rel_err(res) < 1e-8


@pytest.mark.parametrize("model", MODELS)
def test_f32_one_sweep(model):
    """fp32 engine vs fp64 oracle after one sweep.  fp32 rounding flips an accept/reject decision of the PG sampler in about
    one cell per 5e4 (tests/test_gpu_samplers.py); one flipped omega_ij moves b_j by ~omega/S0 ~ 1/N and hence every theta_i
    by a few 1e-6, so the tolerances are absolute: subjects 5e-4 for >= 99.8 % of rows, items 2e-3."""
    N = 2000
    res = pu.run_pair(model, N=N, J=25, nsweeps=1, precision="f32")

    def abs_err(d, o):
        return np.abs(d - o) / np.maximum(np.abs(o), 1.0)

    assert np.mean(abs_err(res["dev_ra"][0, :N], res["orc"]["ra"][0, :N]) > 5e-4) < 2e-3
    assert abs_err(res["dev_ra"][0, N:], res["orc"]["ra"][0, N:]).max() < 2e-3
    if model != "mlirt":
        assert np.mean(abs_err(res["dev_rt"][0, :N], res["orc"]["rt"][0, :N]) > 5e-4) < 2e-3
        assert abs_err(res["dev_rt"][0, N:], res["orc"]["rt"][0, N:]).max() < 2e-3
    assert abs(res["dev_ll"][0] - res["orc"]["ll"][0]) < 1e-4 * abs(res["orc"]["ll"][0])


@pytest.mark.parametrize("precision,tol", [("f64", 1e-8), ("f32", 5e-4)])
@pytest.mark.parametrize("model", MODELS)
def test_teacher_forced(model, precision, tol):
    """Every sweep started from the ORACLE's state of the previous sweep: each conditional is checked on realistic chain
    states without error accumulation.  fp64: everything within 1e-8; fp32: >= 99.8 % of subject draws and all item draws
    within 5e-4 (floor 1e-2 absolute)."""
    T, N, J = 8, 600, 11
    Y, logT, X, init, _ = pu.make_problem(model, N, J)
    L = pu.ge.load_package()._lib
    cov2one = model not in ("latentqr", "latent")
    op = pu.OracleProblem(model, Y, logT, X, init, qRt=0.85, cov2one=cov2one)
    eng = L.Engine(model=pu.MODELS[model], n_item=J, n_subj=N, n_feat=0 if X is None else X.shape[1], n_iter=T, n_chain=1,
                   n_burnin=0, cov2one=int(cov2one), q_rt=0.85, seed=1234, precision={"f32": 0, "f64": 1}[precision], trace_mode=1)
    eng.set_data(Y, logT, X)
    names = dict(theta="theta", a="a", b="b", zeta="zeta", lambda_="lambda_", sig2t="sig2t", beta="beta", sigp="Sigp", rho="rho", nu="nu")
    floor = 1e-6 if precision == "f64" else 1e-2
    for t in range(T):
        st = {k: op.arr[v].copy() for k, v in names.items()}
        if model not in ("crossqr", "latentqr"):
            st.pop("nu")
        if model == "crossqr" and t == 0:
            st.pop("nu")           # constructors leave nu unset; it is drawn first
        if model == "rtirt":
            st["beta"] = st["beta"]
        eng.set_state(**st)
        eng.run(1)
        op.run(1)
        dev = eng.get_state()
        for k, v in names.items():
            if dev[k] is None or (model == "mlirt" and k in ("zeta", "lambda_", "sig2t", "sigp", "rho")):
                continue
            if k == "nu":
                continue           # device nu is already next sweep's draw (fused schedule); it is checked through the next sweep
            e = pu.rel_err(dev[k], op.arr[v], floor)
            if k in ("theta", "zeta") and precision == "f32":
                assert np.mean(e > tol) < 2e-3, (model, t, k, e.max())
            else:
                assert e.max() < tol, (model, t, k, e.max())


@pytest.mark.parametrize("model", ["mlirt", "rtirt", "latentqr", "null", "cross", "latent"])
def test_f32_chain_stays_coupled(model):
    """fp32 engine against the oracle's fp64 chain on the same addressed variates.  Tolerance: the posterior means of
    every item-level parameter within 3 Monte-Carlo standard errors, each standard error taken from the ORACLE chain (sd of its
    post-burn-in draws / sqrt(ESS), the package's split-chain Geyer estimator)."""
    T = 80
    res = pu.run_pair(model, N=1500, J=20, nsweeps=T, precision="f32")
    N = 1500
    ess_rhat = pu.ge.load_package().gibbs.ess_rhat

    def check(dev, orc):
        post = orc[T // 2:]
        se = np.array([post[:, k].std(ddof=1) / np.sqrt(min(max(ess_rhat(post[:, k])[0], 1.0), post.shape[0])) for k in range(post.shape[1])])
        z = np.abs(dev[T // 2:].mean(0) - post.mean(0)) / se
        assert z.max() < 3.0, (model, z.max())
    check(res["dev_ra"][:, N:], res["orc"]["ra"][:, N:])
    if model != "mlirt":
        check(res["dev_rt"][:, N:], res["orc"]["rt"][:, N:])
    assert np.max(np.abs(res["dev_ll"] - res["orc"]["ll"]) / np.abs(res["orc"]["ll"])) < 1e-3


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_bit_reproducible_and_continuable(precision):
    Y, logT, X, init, _ = pu.make_problem("rtirt", 999, 17)
    a = pu.run_device("rtirt", Y, logT, X, init, 10, precision=precision)
    b = pu.run_device("rtirt", Y, logT, X, init, 10, precision=precision)
    assert np.array_equal(a["ra"], b["ra"]) and np.array_equal(a["rt"], b["rt"]) and np.array_equal(a["ll"], b["ll"])
    # run(4) + run(6) == run(10)
    pkg = pu.ge.load_package()
    L = pkg._lib
    eng = L.Engine(model=1, n_item=17, n_subj=999, n_feat=3, n_iter=10, n_chain=1, n_burnin=5, cov2one=1, q_rt=0.85, seed=1234,
                   precision={"f32": 0, "f64": 1}[precision], trace_mode=1)
    eng.set_data(Y, logT, X)
    eng.set_state(**init)
    eng.run(4)
    eng.run(6)
    assert np.array_equal(eng.trace(L.TRACE_RA), a["ra"])
    assert np.array_equal(eng.trace(L.TRACE_LOGLIKE), a["ll"])


def test_interleaved_chain_layout_and_means():
    """nChain > 1 reproduces the reference's interleaved loop: sweep (m, l) -> Post[m, :, l]; Post.mean over m > nBurnin, all l."""
    Y, logT, X, init, _ = pu.make_problem("rtirt", 300, 8)
    one = pu.run_device("rtirt", Y, logT, X, init, 12, precision="f64", n_chain=1, n_burnin=6)
    three = pu.run_device("rtirt", Y, logT, X, init, 12, precision="f64", n_chain=3, n_burnin=2)
    for m in range(4):
        for l in range(3):
            assert np.array_equal(three["ra"][m, :, l], one["ra"][m * 3 + l, :, 0])
    mean = three["engine"].get_mean()
    ref = three["ra"][2:, :300, :].mean(axis=(0, 2))
    assert np.max(np.abs(mean["theta"] - ref)) < 1e-12
    ref_b = three["ra"][2:, 300 + 8:, :].mean(axis=(0, 2))
    assert np.max(np.abs(mean["b"] - ref_b)) < 1e-12


def test_summary_mode_matches_full_mode():
    Y, logT, X, init, _ = pu.make_problem("latentqr", 500, 10)
    full = pu.run_device("latentqr", Y, logT, X, init, 8, precision="f64")
    summ = pu.run_device("latentqr", Y, logT, X, init, 8, precision="f64", trace_full=False)
    assert np.array_equal(full["item"], summ["item"])
    mf, ms = full["engine"].get_mean(), summ["engine"].get_mean()
    for k in ("theta", "zeta", "nu", "a", "beta", "sigp"):
        assert np.array_equal(mf[k], ms[k])
    with pytest.raises(Exception):
        summ["engine"].trace(0)


@pytest.mark.parametrize("model,N,J,F", [("rtirt", 64, 896, 1), ("rtirt", 300, 5, 14), ("latentqr", 300, 5, 14), ("latent", 300, 5, 14),
                                         ("mlirt", 300, 5, 14), ("rtirt", 2, 3, 0), ("crossqr", 3, 2, 0), ("latentqr", 5, 1, 1)])
def test_f64_limits_of_the_engine(model, N, J, F):
    """The documented limits (nItem <= 896, nFeat <= 14) and degenerate sizes (a handful of subjects / one item)."""
    res = pu.run_pair(model, N=N, J=J, nsweeps=3, F=F, precision="f64")
    assert pu.max_rel_err(res, floor=1e-5) < 1e-7


def test_two_engines_in_one_process_do_not_interfere():
    """Distinct handles are independent (include/ertirt.h): interleaved runs of two engines equal their separate runs."""
    Y, logT, X, init, _ = pu.make_problem("rtirt", 400, 8)
    ref = pu.run_device("rtirt", Y, logT, X, init, 6, precision="f64")["ra"]
    L = pu.ge.load_package()._lib
    engs = []
    for _ in range(2):
        e = L.Engine(model=1, n_item=8, n_subj=400, n_feat=3, n_iter=6, n_chain=1, n_burnin=3, cov2one=1, q_rt=0.85, seed=1234, precision=1, trace_mode=1)
        e.set_data(Y, logT, X)
        e.set_state(**init)
        engs.append(e)
    for k in range(3):
        for e in engs:
            e.run(2)
    for e in engs:
        assert np.array_equal(e.trace(L.TRACE_RA), ref)


@pytest.mark.parametrize("model", ["rtirt", "crossqr"])
def test_graph_replay_and_continuation(model):
    """Long enough for the 32-sweep hipGraph path: run(45) + run(40) + run(3) == run(88), bit for bit (the fused kernel's double-buffered
    parameter / counter / statistics blocks are re-based at every erm_run so the captured graph always sees the parity it was built with)."""
    Y, logT, X, init, _ = pu.make_problem(model, 500, 9)
    a = pu.run_device(model, Y, logT, X, init, 88, precision="f32")
    L = pu.ge.load_package()._lib
    eng = L.Engine(model=pu.MODELS[model], n_item=9, n_subj=500, n_feat=0 if X is None else 3, n_iter=88, n_chain=1, n_burnin=44, cov2one=1, q_rt=0.85,
                   seed=1234, precision=0, trace_mode=1)
    eng.set_data(Y, logT, X)
    eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in init.items()})
    for n in (45, 40, 3):
        eng.run(n)
    assert np.array_equal(eng.trace(L.TRACE_RA), a["ra"]) and np.array_equal(eng.trace(L.TRACE_LOGLIKE), a["ll"])
    assert np.array_equal(eng.item_trace(), a["item"])
    # profile mode (bench.py) interleaves event-bracketed single sweeps with the replayed blocks: same chain, and launches were timed
    b = pu.run_device(model, Y, logT, X, init, 88, precision="f32", profile=1)
    assert np.array_equal(b["ra"], a["ra"]) and np.array_equal(b["ll"], a["ll"]) and np.array_equal(b["item"], a["item"])
    assert b["engine"].timing()["pass_launches"] >= 2
