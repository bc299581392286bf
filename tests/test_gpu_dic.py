"""getDic from device-resident state (erm_get_dic / erm_farm_get_dic) against the host evaluation of the same definition
(/root/reference/src/GibbsRtIrt.pl.jl:432-472, src/GibbsRtIrtCross.pl.jl:330-353, src/GibbsRtIrtLatent.pl.jl:342-365: D̂ = -2 logLik(Post.mean),
D̄ = -2 mean(Post.logLike) over all iterations) and against the oracle's log-likelihood at Post.mean, for all seven samplers."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu

CLS = {"mlirt": "GibbsMlIrt", "rtirt": "GibbsRtIrt", "crossqr": "GibbsRtIrtCrossQr", "latentqr": "GibbsRtIrtLatentQr", "null": "GibbsRtIrtNull", "cross": "GibbsRtIrtCross",
       "latent": "GibbsRtIrtLatent"}


def _sampler(pkg, model, N, J, *, precision, nIter=20, nChain=2, seed=11):
    Y, logT, X, init, tp = pu.make_problem(model, N, J, 3, seed=seed)
    Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=3, nIter=nIter, nChain=nChain, qRt=0.85)
    D = pkg.InputData(Y=Y, T=np.exp(logT) if logT is not None else (), X=X if X is not None else np.zeros((N, 3)))
    return getattr(pkg, CLS[model])(Cond, Data=D, precision=precision), (Y, logT, X)


def _oracle_loglik_at_mean(model, M, data):
    Y, logT, X = data
    P = M.Post.mean
    st = dict(theta=P.theta, a=P.a, b=P.b)
    if model != "mlirt":
        st.update(zeta=P.zeta, lam=P.lam, sig2t=P.sig2t, sigp=P.Sigp)
    if P.beta.size and model != "null":
        st["beta"] = P.beta
    if P.rho.size:
        st["rho"] = P.rho
    if P.nu.size:
        st["nu"] = P.nu
    op = pu.OracleProblem(model, Y, logT, X, st, qRt=0.85)
    return op.loglik()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("model", list(CLS))
def test_device_dic_equals_the_host_evaluation_and_the_oracle(model, precision):
    pkg = pu.ge.load_package()
    M, data = _sampler(pkg, model, 300, 12, precision=precision)
    pkg.sample_b(M)
    d = M._engine.dic()
    host = pkg.getDicHost(M)
    Dbar_host = -2.0 * float(np.mean(M.Post.logLike))
    Dhat_host = -2.0 * pkg.getLogLikelihood(M, M.Post.mean)
    # the fp32 engine keeps the centred logT in fp32: its log-likelihood is that of the rounded data set (1e-7 relative per cell)
    tol = 1e-10 if precision == "f64" else 2e-6
    assert abs(d["Dbar"] - Dbar_host) <= 1e-12 * abs(Dbar_host)
    assert abs(d["Dhat"] - Dhat_host) <= tol * abs(Dhat_host)
    assert abs(d["Dhat"] + 2.0 * _oracle_loglik_at_mean(model, M, data)) <= tol * abs(Dhat_host)
    assert abs(d["pD"] - (d["Dbar"] - d["Dhat"])) <= 1e-9 * abs(d["Dbar"]) and abs(d["DIC"] - (d["Dbar"] + d["pD"])) <= 1e-9 * abs(d["Dbar"])
    got = pkg.getDic(M)                       # the public getDic takes the device path
    assert got.DIC == d["DIC"] and got.pD == d["pD"]
    assert abs(got.DIC - host.DIC) <= 50 * tol * abs(host.DIC)
    M.close()


def test_dic_needs_post_burnin_rows_and_a_summary_engine_suffices():
    """erm_get_dic reads the logLike rows, the running sums and the item trace: ERM_TRACE_SUMMARY engines have all of them; before any post-burn-in row
    it refuses."""
    pkg = pu.ge.load_package()
    L = pkg._lib
    Y, logT, X, init, tp = pu.make_problem("rtirt", 200, 10, 3, seed=3)
    out = {}
    for mode in ("summary", "full"):
        eng = L.Engine(model=L.MODEL_RTIRT, n_item=10, n_subj=200, n_feat=3, n_iter=12, n_chain=1, n_burnin=6, cov2one=1, q_rt=0.5, seed=5, precision=L.PREC_F64,
                       trace_mode=L.TRACE_SUMMARY if mode == "summary" else L.TRACE_FULL)
        eng.set_data(Y, logT, X)
        eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in init.items()})
        with pytest.raises(L.ErmError, match="post-burn-in"):
            eng.dic()
        eng.run(4)
        with pytest.raises(L.ErmError, match="post-burn-in"):
            eng.dic()
        eng.run(8)
        out[mode] = eng.dic()
        eng.close()
    assert out["summary"] == out["full"]


def test_farm_dic_is_the_dic_of_the_pooled_chains():
    """erm_farm_get_dic: D̄ over the logLike rows of every chain, D̂ at the joint Post.mean (the vector erm_farm_get_mean reduces), here with both chains on
    device 0 and the reduction forced through the library's RCCL communicator."""
    pkg = pu.ge.load_package()
    L = pkg._lib
    M, data = _sampler(pkg, "rtirt", 300, 12, precision="f64", nIter=16, nChain=2)
    M.engine_opts["flags"] = L.FLAG_FARM_FORCE_RCCL
    pkg.sample_b(M, devices=[0, 0])
    d = M.farm.dic()
    assert M.farm.used_rccl
    Dbar = -2.0 * float(np.mean(M.Post.logLike))
    Dhat = -2.0 * pkg.getLogLikelihood(M, M.Post.mean)
    assert abs(d["Dbar"] - Dbar) <= 1e-12 * abs(Dbar) and abs(d["Dhat"] - Dhat) <= 1e-10 * abs(Dhat)
    assert pkg.getDic(M).DIC == d["DIC"]
    M.close()


@pytest.mark.parametrize("model", ["rtirt", "crossqr"])
def test_dic_at_baseline_size_matches_the_host_evaluation(model):
    """BASELINE.json's size (100 000 x 50): the device DIC of a summary-trace engine against the numpy evaluation of getLogLikelihood at the pulled Post.mean
    and the pulled logLike rows -- for GibbsRtIrtCrossQr that includes the 5 000 000 means of nu the device evaluation never moves."""
    pkg = pu.ge.load_package()
    L = pkg._lib
    N, J, rows, burn = 100_000, 50, 12, 4
    Y, logT, X, init, tp = pu.make_problem(model, N, J, 3, seed=21)
    r = pu.run_device(model, Y, logT, X, init, rows, precision="f64", trace_full=False, n_burnin=burn)
    eng = r["engine"]
    d = eng.dic()
    m = eng.get_mean()
    Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=3, nIter=rows, nChain=1, qRt=0.85)
    M = getattr(pkg, CLS[model])(Cond, Data=pkg.InputData(Y=Y, T=np.exp(logT), X=X if X is not None else np.zeros((N, 3))))
    P = pkg.InputPara(theta=m["theta"], a=m["a"], b=m["b"], zeta=m["zeta"], lam=m["lambda_"], sig2t=m["sig2t"], Sigp=m["sigp"])
    for k_src, k_dst in (("beta", "beta"), ("rho", "rho"), ("nu", "nu")):
        if m.get(k_src) is not None:
            setattr(P, k_dst, m[k_src])
    Dhat = -2.0 * pkg.getLogLikelihood(M, P)
    Dbar = -2.0 * float(np.mean(r["ll"]))
    assert abs(d["Dhat"] - Dhat) <= 1e-10 * abs(Dhat) and abs(d["Dbar"] - Dbar) <= 1e-12 * abs(Dbar)
    eng.close()
