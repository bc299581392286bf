"""BASELINE.json's full sizes (nSubj=100000, nItem=50).  The oracle needs ~0.5 s per sweep here, so it checks two sweeps of the
fp64 engine elementwise; everything else is checked through size-independent properties: bit-reproducibility, run(a)+run(b) ==
run(a+b), geometry independence of the fp64 draws, finite traces, and parameter recovery against the generating truth."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu
N, J = 100_000, 50


@pytest.fixture(scope="module")
def rtirt():
    return pu.make_problem("rtirt", N, J, 3, seed=1234, qRt=0.5)


def test_fullsize_f64_two_sweeps_match_oracle(rtirt):
    Y, logT, X, init, _ = rtirt
    dev = pu.run_device("rtirt", Y, logT, X, init, 2, precision="f64", qRt=0.5)
    op = pu.OracleProblem("rtirt", Y, logT, X, init, qRt=0.5)
    tr = op.run(2)
    assert pu.rel_err(dev["ra"][:, :, 0], tr["ra"]).max() < 1e-8
    assert pu.rel_err(dev["rt"][:, :, 0], tr["rt"]).max() < 1e-8
    assert pu.rel_err(dev["qr"][:, :, 0], tr["qr"]).max() < 1e-8
    assert pu.rel_err(dev["ll"][:, 0, 0], tr["ll"]).max() < 1e-9


def test_fullsize_f32_reproducible_continuable_and_recovers_truth(rtirt):
    Y, logT, X, init, tp = rtirt
    T = 120
    a = pu.run_device("rtirt", Y, logT, X, init, T, precision="f32", qRt=0.5, trace_full=False)
    b = pu.run_device("rtirt", Y, logT, X, init, T, precision="f32", qRt=0.5, trace_full=False, lanes_per_row=16, block_threads=512)
    assert np.array_equal(a["item"], pu.run_device("rtirt", Y, logT, X, init, T, precision="f32", qRt=0.5, trace_full=False)["item"])
    assert np.all(np.isfinite(a["item"])) and np.all(np.isfinite(a["ll"]))
    # a different launch geometry changes only summation order (which subjects share a 4-cell fp32 partial sum of the column phase): over the first ten
    # sweeps the item traces stay within fp32-mode tolerance at this N (measured 2.6e-4 with round-robin batches, 0.6e-4 with per-wave slices)
    assert np.max(np.abs(a["item"][:10] - b["item"][:10])) < 5e-4
    m = a["engine"].get_mean()
    assert np.sqrt(np.mean((m["a"] - tp.a) ** 2)) < 0.03 and np.sqrt(np.mean((m["b"] - tp.b) ** 2)) < 0.03
    # the generator truncates logT at 0 (src/SimTools.jl:169), which shrinks the residual variance the model sees by ~7 %
    assert np.max(np.abs(m["sig2t"] / tp.sig2t - 1)) < 0.12 and np.corrcoef(m["sig2t"], tp.sig2t)[0, 1] > 0.99
    assert np.corrcoef(m["theta"], tp.theta)[0, 1] > 0.95 and np.corrcoef(m["zeta"], tp.zeta)[0, 1] > 0.99
    assert a["ll"][-1, 0, 0] > a["ll"][0, 0, 0]


@pytest.mark.parametrize("model", ["mlirt", "latentqr", "crossqr"])
def test_fullsize_other_models_run_clean(model):
    Y, logT, X, init, tp = pu.make_problem(model, N, J, 3, seed=77, qRt=0.85)
    T = 40
    r = pu.run_device(model, Y, logT, X, init, T, precision="f32", qRt=0.85, trace_full=False)
    assert np.all(np.isfinite(r["item"])) and np.all(np.isfinite(r["ll"]))
    assert np.array_equal(r["item"], pu.run_device(model, Y, logT, X, init, T, precision="f32", qRt=0.85, trace_full=False)["item"])
    m = r["engine"].get_mean()
    assert np.sqrt(np.mean((m["b"] - tp.b) ** 2)) < 0.1
    assert np.corrcoef(m["theta"], tp.theta)[0, 1] > 0.9


# ------------------------------------------------------------------------------------------------------------------------------
# fp64 (the product default, the bench headline) at PRODUCTION geometry for every kernel instantiation.  Each model is its own template
# instantiation of pass_kernel<MODEL, double, ...> (LatentQr is even compiled for a different workgroup size); the small-data rule of the planner
# shrinks workgroups to 256 threads below ~6 000 subjects, so only sizes like these exercise the 1024-thread (768 for LatentQr) kernels the
# BASELINE.md numbers come from.  Reference loops: /root/reference/src/GibbsRtIrt.pl.jl:210-257 (MlIrt), :367-426 (Null),
# src/GibbsRtIrtLatent.pl.jl:271-337 (LatentQr), :168-233 (Latent), src/GibbsRtIrtCross.pl.jl:265-325 (CrossQr), :176-235 (Cross).
# ------------------------------------------------------------------------------------------------------------------------------
OTHERS = ["mlirt", "latentqr", "crossqr", "null", "cross", "latent"]


def _oracle_sweeps(model, Y, logT, X, init, T, qRt):
    pu.oracle().orc_set_threads(16)                      # OpenMP mode: bit-identical to the single-thread run (test_oracle_sweeps.py)
    try:
        op = pu.OracleProblem(model, Y, logT, X, init, qRt=qRt, cov2one=(model not in ("latentqr", "latent")))
        return op, op.run(T, with_nu=(model in ("latentqr", "crossqr")))
    finally:
        pu.oracle().orc_set_threads(1)


def _assert_traces(model, dev, tr, tol=1e-8):
    """Every trace column of every sweep within `tol` -- except GibbsRtIrtCrossQr's sweeps after the first: its chain is chaotic (the 1/nu
    weights pin zeta_i to its smallest-nu residual, a 1 ulp difference grows to 1e-2 within 12 sweeps), and over 100 000 subjects the largest
    deviation of sweep 2 already reaches ~5e-8; sweep 1 is held to `tol`, later free-running sweeps to 1e-6, and the model is checked
    teacher-forced besides (test_fullsize_f64_crossqr_teacher_forced_sweep)."""
    rows = dev["ra"].shape[0]
    tols = np.array([tol if (model != "crossqr" or t == 0) else 1e-6 for t in range(rows)])[:, None]
    assert np.all(pu.rel_err(dev["ra"][:, :, 0], tr["ra"]) < tols)
    if model != "mlirt":
        assert np.all(pu.rel_err(dev["rt"][:, :, 0], tr["rt"]) < tols)
    assert np.all(pu.rel_err(dev["qr"][:, :, 0], tr["qr"]) < tols)          # LatentQr: every nu_i; CrossQr: every nu_ij (5e6 per sweep)
    assert np.all(pu.rel_err(dev["ll"][:, 0, 0], tr["ll"]) < 0.1 * tols[:, 0])


@pytest.mark.parametrize("model", OTHERS)
def test_fullsize_f64_every_model_matches_oracle_at_production_geometry(model):
    """100 000 x 50, qRt = 0.85, default geometry, fp64: two free-running sweeps elementwise against the oracle (CrossQr including every nu), then
    bit-reproducibility and run(a) + run(b) == run(a + b)."""
    Y, logT, X, init, _ = pu.make_problem(model, N, J, 3, seed=77, qRt=0.85)
    dev = pu.run_device(model, Y, logT, X, init, 2, precision="f64", qRt=0.85, n_burnin=0)
    tm = dev["engine"].timing()
    assert tm["block_threads"] == (768 if model == "latentqr" else 1024) and tm["grid_blocks"] == tm["cu_count"]       # one full-size workgroup per CU
    _, tr = _oracle_sweeps(model, Y, logT, X, init, 2, 0.85)
    _assert_traces(model, dev, tr)
    del dev
    T = 7
    a = pu.run_device(model, Y, logT, X, init, T, precision="f64", qRt=0.85, trace_full=False)
    b = pu.run_device(model, Y, logT, X, init, T, precision="f64", qRt=0.85, trace_full=False)
    assert np.array_equal(a["item"], b["item"]) and np.array_equal(a["ll"], b["ll"])
    assert np.all(np.isfinite(a["item"])) and np.all(np.isfinite(a["ll"]))
    L = pu.ge.load_package()._lib
    eng = L.Engine(model=pu.MODELS[model], n_item=J, n_subj=N, n_feat=0 if X is None else 3, n_iter=T, n_chain=1, n_burnin=T // 2,
                   cov2one=int(model not in ("latentqr", "latent")), q_rt=0.85, seed=1234, precision=1, trace_mode=0)
    eng.set_data(Y, logT, X)
    eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in init.items()})
    for n in (3, 1, 3):
        eng.run(n)
    assert np.array_equal(eng.item_trace(), a["item"]) and np.array_equal(eng.trace(L.TRACE_LOGLIKE), a["ll"])
    ma, me = a["engine"].get_mean(), eng.get_mean()
    for k in ("theta", "zeta", "nu"):
        if ma[k] is not None:
            assert np.array_equal(ma[k], me[k]), k


def test_fullsize_f64_crossqr_teacher_forced_sweep():
    """GibbsRtIrtCrossQr is chaotic (1 ulp -> 1e-2 within 12 sweeps), so beyond its free-running sweeps it is checked teacher-forced:
    sweep 3 started from the ORACLE's state after sweep 2, every conditional on a realistic chain state at full size."""
    model = "crossqr"
    Y, logT, X, init, _ = pu.make_problem(model, N, J, 3, seed=78, qRt=0.85)
    op, _ = _oracle_sweeps(model, Y, logT, X, init, 2, 0.85)
    L = pu.ge.load_package()._lib
    names = dict(theta="theta", a="a", b="b", zeta="zeta", lambda_="lambda_", sig2t="sig2t", sigp="Sigp", rho="rho", nu="nu")
    # the oracle's sweep counter stands at 2: the device must draw sweep 3's variates -- its own counter is advanced by two throw-away sweeps first
    eng2 = L.Engine(model=pu.MODELS[model], n_item=J, n_subj=N, n_feat=0, n_iter=3, n_chain=1, n_burnin=0, cov2one=1, q_rt=0.85, seed=1234, precision=1, trace_mode=0)
    eng2.set_data(Y, logT, X)
    eng2.set_state(**{("lambda_" if k == "lam" else k): v for k, v in init.items()})
    eng2.run(2)
    st = {k: op.arr[v].copy() for k, v in names.items()}
    eng2.set_state(**st)
    eng2.run(1)
    pu.oracle().orc_set_threads(16)
    try:
        op.run(1)
    finally:
        pu.oracle().orc_set_threads(1)
    dev = eng2.get_state()
    for k, v in names.items():
        if k == "nu":
            continue               # the device's nu is already the next sweep's draw (fused schedule)
        assert pu.rel_err(dev[k], op.arr[v], 1e-6).max() < 1e-8, k


@pytest.mark.parametrize("model", ["rtirt"] + OTHERS)
def test_midsize_f64_one_sweep_matches_oracle_in_full_size_workgroups(model):
    """30 000 x 50: still one 1024-thread (768) workgroup per CU, a third of the subjects per wave -- another point of the same kernels."""
    n = 30_000
    Y, logT, X, init, _ = pu.make_problem(model, n, J, 3, seed=79, qRt=0.85)
    dev = pu.run_device(model, Y, logT, X, init, 1, precision="f64", qRt=0.85, n_burnin=0)
    assert dev["engine"].timing()["block_threads"] == (768 if model == "latentqr" else 1024)
    _, tr = _oracle_sweeps(model, Y, logT, X, init, 1, 0.85)
    _assert_traces(model, dev, tr)


# ------------------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: GibbsRtIrt nSubj = 500000, nItem = 100 (one chain's load on one GPU).  The working set (Y 50 MB, logT and
# omega 200 / 400 MB each) no longer fits the 256 MiB Infinity Cache, a workgroup owns ~1950 subjects x 100 items and the LDS layout
# grows with nItem -- a regime of its own for the engine.
# ------------------------------------------------------------------------------------------------------------------------------
N4, J4 = 500_000, 100


@pytest.fixture(scope="module")
def rtirt4():
    return pu.make_problem("rtirt", N4, J4, 3, seed=4321, qRt=0.5)


def test_configs4_f64_one_sweep_matches_oracle(rtirt4):
    Y, logT, X, init, _ = rtirt4
    dev = pu.run_device("rtirt", Y, logT, X, init, 1, precision="f64", qRt=0.5, n_burnin=0)
    pu.oracle().orc_set_threads(16)                      # the oracle's OpenMP mode: bit-identical to its single-thread run (test_oracle_sweeps.py)
    try:
        tr = pu.OracleProblem("rtirt", Y, logT, X, init, qRt=0.5).run(1)
    finally:
        pu.oracle().orc_set_threads(1)
    assert pu.rel_err(dev["ra"][:, :, 0], tr["ra"]).max() < 1e-8
    assert pu.rel_err(dev["rt"][:, :, 0], tr["rt"]).max() < 1e-8
    assert pu.rel_err(dev["qr"][:, :, 0], tr["qr"]).max() < 1e-8
    assert pu.rel_err(dev["ll"][:, 0, 0], tr["ll"]).max() < 1e-9


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_configs4_reproducible_continuable_geometry_invariant_and_recovers_truth(rtirt4, precision):
    Y, logT, X, init, tp = rtirt4
    T = 60
    a = pu.run_device("rtirt", Y, logT, X, init, T, precision=precision, qRt=0.5, trace_full=False)
    assert np.all(np.isfinite(a["item"])) and np.all(np.isfinite(a["ll"]))
    b = pu.run_device("rtirt", Y, logT, X, init, T, precision=precision, qRt=0.5, trace_full=False)
    assert np.array_equal(a["item"], b["item"]) and np.array_equal(a["ll"], b["ll"])              # bit-reproducible
    # run(x) + run(y) == run(x + y)
    L = pu.ge.load_package()._lib
    eng = L.Engine(model=pu.MODELS["rtirt"], n_item=J4, n_subj=N4, n_feat=3, n_iter=T, n_chain=1, n_burnin=T // 2, cov2one=1, q_rt=0.5, seed=1234,
                   precision={"f32": 0, "f64": 1}[precision], trace_mode=0)
    eng.set_data(Y, logT, X)
    eng.set_state(**init)
    for n in (T // 3, T - T // 3 - 5, 5):
        eng.run(n)
    assert np.array_equal(eng.item_trace(), a["item"]) and np.array_equal(eng.trace(L.TRACE_LOGLIKE), a["ll"])
    ma, me = a["engine"].get_mean(), eng.get_mean()
    assert np.array_equal(ma["theta"], me["theta"]) and np.array_equal(ma["zeta"], me["zeta"])
    del eng, b
    # another launch geometry changes only the summation order of the statistics
    geo = dict(lanes_per_row=16, block_threads=512, grid_blocks=300) if precision == "f32" else dict(lanes_per_row=16, block_threads=256, grid_blocks=1500)
    g = pu.run_device("rtirt", Y, logT, X, init, 8, precision=precision, qRt=0.5, trace_full=False, **geo)
    tol = 2e-4 if precision == "f32" else 1e-9
    assert np.max(np.abs(a["item"][:8] - g["item"]) / np.maximum(np.abs(a["item"][:8]), 1.0)) < tol
    # recovery of the generating values (README.md:61-77 style).  After 60 sweeps from a = 1, b = 0 the common scale of (a, theta) is
    # still drifting towards its stationary value at this N (the posterior is narrow, the scale moves ~N^-1/2 per sweep), so the
    # discriminations are checked up to that common factor; everything else is checked in absolute terms.
    assert np.corrcoef(ma["a"], tp.a)[0, 1] > 0.995 and np.corrcoef(ma["b"], tp.b)[0, 1] > 0.995
    ratio = ma["a"] / tp.a
    assert ratio.std() / ratio.mean() < 0.05 and 0.6 < ratio.mean() < 1.4
    # (lambda, zeta) share a location that is still settling too; the generator's truncation of logT at 0 (src/SimTools.jl:169) shifts each lambda_j a little
    assert np.std(ma["lambda_"] - tp.lam) < 0.03 and abs(np.mean(ma["lambda_"] - tp.lam)) < 0.15 and np.corrcoef(ma["lambda_"], tp.lam)[0, 1] > 0.995
    assert np.max(np.abs(ma["sig2t"] / tp.sig2t - 1)) < 0.12 and np.corrcoef(ma["sig2t"], tp.sig2t)[0, 1] > 0.99
    assert np.corrcoef(ma["theta"], tp.theta)[0, 1] > 0.95 and np.corrcoef(ma["zeta"], tp.zeta)[0, 1] > 0.99


def test_fullsize_f32_chain_within_three_mc_standard_errors_of_the_oracle_chain(rtirt):
    """The fp32 fast mode against the oracle's fp64 chain AT A BASELINE SIZE (configs[2], 100 000 x 50), same addressed variates: posterior
    means of every item-level parameter within 3 Monte-Carlo standard errors, the standard errors from the ORACLE chain (sd of its post-burn-in
    draws / sqrt(ESS)) -- SURVEY.md 8(c)(4).  The oracle runs its OpenMP mode (bit-identical to one thread): 80 sweeps in a few seconds."""
    Y, logT, X, init, _ = rtirt
    T = 80
    dev = pu.run_device("rtirt", Y, logT, X, init, T, precision="f32", qRt=0.5, trace_full=False)
    pu.oracle().orc_set_threads(16)
    try:
        tr = pu.OracleProblem("rtirt", Y, logT, X, init, qRt=0.5).run(T)
    finally:
        pu.oracle().orc_set_threads(1)
    ess_rhat = pu.ge.load_package().gibbs.ess_rhat
    orc_item = np.column_stack([tr["ra"][:, N:], tr["rt"][:, N:]])          # a, b | lambda, sig2t
    dev_item = dev["item"][:, :4 * J]
    post = orc_item[T // 2:]
    se = np.array([post[:, k].std(ddof=1) / np.sqrt(min(max(ess_rhat(post[:, k])[0], 1.0), post.shape[0])) for k in range(post.shape[1])])
    z = np.abs(dev_item[T // 2:].mean(0) - post.mean(0)) / se
    assert z.max() < 3.0, z.max()
    assert np.max(np.abs(dev["ll"][:, 0, 0] - tr["ll"]) / np.abs(tr["ll"])) < 1e-4
