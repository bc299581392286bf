"""Synthetic truth / data generators restating the *distributions* of
/root/reference/src/SimTools.jl:74-368 (setTruePara*, setData*) with numpy's own generator (Julia's Random.seed! streams
cannot be reproduced outside Julia).  They define the benchmark and parity-test inputs; recovery metrics getRmse / getBias
follow src/SimTools.jl:42-45.
"""
from __future__ import annotations

import numpy as np

from .base import InputData, InputPara, SimConditions


def _rng(seed):
    return seed if isinstance(seed, np.random.Generator) else np.random.default_rng(seed)


def _truncnorm(g, mu, sd, lo, hi, size):
    """Truncated(Normal(mu, sd), lo, hi) by inverse cdf, evaluated through whichever tail keeps it well conditioned
    (plain rejection never terminates when the truncation region is far in a tail, e.g. logT > 0 with lambda - zeta << 0)."""
    from scipy.special import ndtr, ndtri
    mu = np.broadcast_to(np.asarray(mu, dtype=np.float64), size)
    sd = np.broadcast_to(np.asarray(sd, dtype=np.float64), size)
    a, b = (lo - mu) / sd, (hi - mu) / sd
    u = g.random(size)
    upper = a > 0                                   # work with survival probabilities when the region is in the upper tail
    pa, pb = np.where(upper, ndtr(-a), ndtr(a)), np.where(upper, ndtr(-b), ndtr(b))
    q = pa + u * (pb - pa)
    z = np.where(upper, -ndtri(q), ndtri(q))
    return np.clip(mu + sd * z, lo, hi)


def setTrueParaRtIrt(Cond: SimConditions, *, trueStdRa=1.0, trueStdRt=1.0, trueCorr=0.0, seed=1234):
    """src/SimTools.jl:74-95"""
    g = _rng(seed)
    P = InputPara()
    P.a = _truncnorm(g, 1.0, 0.2, 0.0, np.inf, (Cond.nItem,))
    P.b = g.normal(0.0, 0.5, Cond.nItem)
    P.lam = _truncnorm(g, 4.0, 0.2, 0.0, np.inf, (Cond.nItem,))
    P.sig2t = np.exp(g.normal(np.log(0.3), 0.2, Cond.nItem))
    sd = np.diag([trueStdRa, trueStdRt])
    P.Sigp = sd @ np.array([[1.0, trueCorr], [trueCorr, 1.0]]) @ sd
    P.beta = g.standard_normal((Cond.nFeat, 2))
    return P


def setTrueParaMlIrt(Cond: SimConditions, *, seed=1234):
    """src/SimTools.jl:100-112"""
    g = _rng(seed)
    P = InputPara()
    P.a = _truncnorm(g, 1.0, 0.2, 0.0, np.inf, (Cond.nItem,))
    P.b = g.normal(0.0, 0.5, Cond.nItem)
    P.beta = g.standard_normal((Cond.nFeat, 1))
    return P


def _bernoulli_logit(g, eta):
    return (g.random(eta.shape) < 1.0 / (1.0 + np.exp(-eta))).astype(np.uint8)


def setDataRtIrt(Cond: SimConditions, truePara: InputPara, *, seed=4321):
    """src/SimTools.jl:149-178"""
    g = _rng(seed)
    X = g.standard_normal((Cond.nSubj, Cond.nFeat))
    L = np.linalg.cholesky(np.asarray(truePara.Sigp).reshape(2, 2))
    subj = X @ np.asarray(truePara.beta).reshape(Cond.nFeat, 2) + g.standard_normal((Cond.nSubj, 2)) @ L.T
    truePara.theta, truePara.zeta = subj[:, 0], subj[:, 1]
    Y = _bernoulli_logit(g, truePara.a[None, :] * (truePara.theta[:, None] - truePara.b[None, :]))
    mut = truePara.lam[None, :] - truePara.zeta[:, None]
    logT = _truncnorm(g, mut, np.sqrt(truePara.sig2t)[None, :], 0.0, np.inf, mut.shape)   # :169 truncates logT at 0
    return InputData(Y=Y, X=X, T=np.exp(logT))


def setDataMlIrt(Cond: SimConditions, truePara: InputPara, *, seed=4321):
    """src/SimTools.jl:349-368"""
    g = _rng(seed)
    X = np.empty((Cond.nSubj, Cond.nFeat))
    X[:, 0] = g.random(Cond.nSubj) < 0.5
    X[:, 1:] = g.standard_normal((Cond.nSubj, Cond.nFeat - 1))
    truePara.theta = X @ np.asarray(truePara.beta).reshape(Cond.nFeat) + g.standard_normal(Cond.nSubj)
    Y = _bernoulli_logit(g, truePara.a[None, :] * (truePara.theta[:, None] - truePara.b[None, :]))
    return InputData(Y=Y, X=X)


def setTrueParaRtIrtCross(Cond: SimConditions, *, trueStdRa=1.0, trueStdRt=1.0, seed=1234):
    """src/SimTools.jl:188-212"""
    g = _rng(seed)
    P = InputPara()
    P.a = _truncnorm(g, 1.0, 0.2, 0.0, np.inf, (Cond.nItem,))
    P.b = g.normal(0.0, 0.5, Cond.nItem)
    P.lam = _truncnorm(g, 3.0, 0.5, 0.0, np.inf, (Cond.nItem,))
    P.sig2t = np.exp(g.normal(np.log(0.3), 0.2, Cond.nItem))
    P.Sigp = np.diag([trueStdRa ** 2, trueStdRt ** 2])
    P.rho = g.normal(0.0, 0.2, Cond.nItem)
    return P


def _noise(g, type, size, sd_norm):
    if type == "norm":
        return g.normal(0.0, sd_norm, size)
    if type == "tail":
        return g.standard_t(5, size)
    if type == "skew":
        return g.gamma(0.5, 1.0, size) - 1.0
    raise ValueError("type must be 'norm', 'tail' or 'skew'")


def setDataRtIrtCross(Cond: SimConditions, truePara: InputPara, *, type="norm", seed=4321):
    """src/SimTools.jl:222-255"""
    g = _rng(seed)
    L = np.linalg.cholesky(np.asarray(truePara.Sigp).reshape(2, 2))
    subj = g.standard_normal((Cond.nSubj, 2)) @ L.T
    truePara.theta, truePara.zeta = subj[:, 0], subj[:, 1]
    Y = _bernoulli_logit(g, truePara.a[None, :] * (truePara.theta[:, None] - truePara.b[None, :]))
    mut = truePara.lam[None, :] - truePara.zeta[:, None] - truePara.theta[:, None] * truePara.rho[None, :]
    logT = mut + _noise(g, type, mut.shape, 0.3)
    return InputData(Y=Y, T=np.exp(logT))


def setTrueParaRtIrtLatent(Cond: SimConditions, *, trueStdRa=1.0, trueStdRt=1.0, seed=1234):
    """src/SimTools.jl:260-293 (sigma2_t is not part of the truth for this family)"""
    g = _rng(seed)
    P = InputPara()
    P.a = _truncnorm(g, 1.0, 0.2, 0.0, np.inf, (Cond.nItem,))
    P.b = g.normal(0.0, 0.5, Cond.nItem)
    P.lam = _truncnorm(g, 3.0, 0.2, 0.0, np.inf, (Cond.nItem,))
    P.Sigp = np.diag([trueStdRa ** 2, trueStdRt ** 2])
    rho = _truncnorm(g, 0.0, 0.5, -1.0, 1.0, (1,))
    P.beta = np.concatenate([g.normal(0.0, 0.5, Cond.nFeat), rho])
    return P


def setDataRtIrtLatent(Cond: SimConditions, truePara: InputPara, *, type="norm", seed=4321):
    """src/SimTools.jl:304-343"""
    g = _rng(seed)
    truePara.theta = g.standard_normal(Cond.nSubj)
    X = g.standard_normal((Cond.nSubj, Cond.nFeat))
    x = np.column_stack([X, truePara.theta])
    truePara.zeta = x @ truePara.beta + _noise(g, type, Cond.nSubj, 0.3)
    Y = _bernoulli_logit(g, truePara.a[None, :] * (truePara.theta[:, None] - truePara.b[None, :]))
    logT = truePara.lam[None, :] - truePara.zeta[:, None] + g.standard_normal((Cond.nSubj, Cond.nItem))
    return InputData(Y=Y, T=np.exp(logT), X=X)


def getRmse(a, b):
    """src/SimTools.jl:42"""
    return float(np.sqrt(np.mean((np.asarray(a) - np.asarray(b)) ** 2)))


def getBias(a, b):
    """src/SimTools.jl:43"""
    return float(np.mean(np.asarray(a) - np.asarray(b)))


def setDataRtIrtNull(Cond: SimConditions, truePara: InputPara, *, seed=4321):
    """src/SimTools.jl:117-144: (theta, zeta) ~ N2(0, Sigp), no covariates"""
    g = _rng(seed)
    L = np.linalg.cholesky(np.asarray(truePara.Sigp).reshape(2, 2))
    subj = g.standard_normal((Cond.nSubj, 2)) @ L.T
    truePara.theta, truePara.zeta = subj[:, 0], subj[:, 1]
    Y = _bernoulli_logit(g, truePara.a[None, :] * (truePara.theta[:, None] - truePara.b[None, :]))
    mut = truePara.lam[None, :] - truePara.zeta[:, None]
    logT = _truncnorm(g, mut, np.sqrt(truePara.sig2t)[None, :], 0.0, np.inf, mut.shape)
    return InputData(Y=Y, T=np.exp(logT))


# ------------------------------------------------------------------------------------------------------------------
# The callers of sample! in a simulation study (src/SimTools.jl:389-443, 457-552).  Host-side bookkeeping around the GPU path.
# ------------------------------------------------------------------------------------------------------------------
_ALIASES = {"λ": "lam", "σ²t": "sig2t", "ρ": "rho", "β": "beta", "Σp": "Sigp", "θ": "theta", "ζ": "zeta", "ν": "nu"}


def _field(obj, name):
    return np.asarray(getattr(obj, _ALIASES.get(name, name)), dtype=np.float64)


_STATE_FIELD = {"a": "a", "b": "b", "lam": "lambda_", "sig2t": "sig2t", "rho": "rho", "beta": "beta", "Sigp": "sigp", "theta": "theta", "zeta": "zeta", "nu": "nu"}


def _device_generator(funcData, funcGibbs):
    """True when funcData is one of this module's own generators and it is the one the sampler's device generator restates (erm_simulate_data picks the
    generator by model: src/SimTools.jl:117-368), so that a replication's data set can be made on the device."""
    from . import _lib
    pairs = {setDataMlIrt: (_lib.MODEL_MLIRT,), setDataRtIrt: (_lib.MODEL_RTIRT,), setDataRtIrtNull: (_lib.MODEL_NULL,),
             setDataRtIrtCross: (_lib.MODEL_CROSS, _lib.MODEL_CROSSQR), setDataRtIrtLatent: (_lib.MODEL_LATENT, _lib.MODEL_LATENTQR)}
    return funcData in pairs and getattr(funcGibbs, "_model", None) in pairs[funcData]


def runSimulation(Cond: SimConditions, truePara: InputPara, *, Para=("a", "b", "λ", "σ²t"), funcData=None, funcGibbs=None, typeName="norm",
                  seed=4321, on_device=None, **gibbs_opts):
    """src/SimTools.jl:457-495: Cond.nRep replications of  data <- funcData(Cond, truePara); MCMC <- funcGibbs(Cond; truePara, Data);
    sample!(MCMC);  keeping Post.mean of the parameters in `Para`, the DIC and checkConvergence's summary of every replication.
    funcData / funcGibbs are callables (the reference looks their names up in Main); defaults setDataRtIrt / GibbsRtIrt.
    Returns {"True": {par: vec}, 1: {par: vec, "Dic": [dic], "Diag": {...}, "Seconds": s}, ..., nRep: {...}} like the reference's Dict.

    on_device (default: whenever funcData is the generator the sampler's model has on the device): the whole study stays on the GPU.  ONE
    sampler and ONE engine serve all replications of the condition; a replication generates its data set there (erm_simulate_data: no host
    generation, no upload), re-seeds the chain (erm_set_seed), samples, and returns summaries only -- the requested Post.mean fields
    (erm_get_mean with only those pointers set), the DIC (erm_get_dic: one evaluation pass at Post.mean on the device) and checkConvergence's
    eight counters (erm_get_convergence).  Neither the data set, nor a trace, nor (unless `Para` names theta / zeta / nu) anything N-wide
    crosses the boundary.  on_device=False is the older host path: numpy generation, upload, a fresh engine per replication."""
    import copy
    import inspect
    import time
    from . import gibbs
    funcData = funcData or setDataRtIrt
    funcGibbs = funcGibbs or gibbs.GibbsRtIrt
    Run = {"True": {p: _field(truePara, p).reshape(-1, order="F") for p in Para}}
    if on_device is None:
        on_device = _device_generator(funcData, funcGibbs)
    elif on_device and not _device_generator(funcData, funcGibbs):
        raise ValueError("on_device=True needs funcData to be the generator of funcGibbs' model (setDataRtIrt for GibbsRtIrt, ...)")
    if on_device:
        MCMC = funcGibbs(Cond, truePara=truePara, seed=seed + 1, **gibbs_opts)
        which = sorted({_STATE_FIELD[_ALIASES.get(p, p)] for p in Para})
        try:
            for run in range(1, Cond.nRep + 1):
                t0 = time.perf_counter()
                MCMC.seed = seed + run
                MCMC.setInitialValues()                  # the constructor's draw of every replication (src/GibbsRtIrt.pl.jl:100-102)
                tp = copy.copy(truePara)
                gibbs.simulateData(MCMC, tp, type=typeName, seed=int(np.random.SeedSequence([seed, run]).generate_state(1, dtype=np.uint64)[0]),
                                   pull=False, pull_truth=False)
                MCMC._engine.set_seed(seed + run)
                gibbs.sample_b(MCMC, fill=False)
                eng = MCMC._engine
                m = eng.get_mean(which)
                Post = {p: np.asarray(m[_STATE_FIELD[_ALIASES.get(p, p)]], dtype=np.float64).reshape(-1) for p in Para}
                Post["Dic"] = [eng.dic()["DIC"]]
                Post["Diag"] = gibbs.checkConvergence(MCMC, detail=False) if MCMC.trace == "full" else None
                Post["Seconds"] = time.perf_counter() - t0
                Run[run] = Post
        finally:
            MCMC.close()
        return Run
    takes_type = "type" in inspect.signature(funcData).parameters
    for run in range(1, Cond.nRep + 1):
        t0 = time.perf_counter()
        kw = dict(seed=np.random.SeedSequence([seed, run]))
        if takes_type:
            kw["type"] = typeName
        Data = funcData(Cond, truePara, **kw)
        MCMC = funcGibbs(Cond, truePara=truePara, Data=Data, seed=seed + run, **gibbs_opts)
        try:
            gibbs.sample_b(MCMC)
            Post = {p: _field(MCMC.Post.mean, p).reshape(-1, order="F") for p in Para}
            Post["Dic"] = [gibbs.getDic(MCMC).DIC]
            Post["Diag"] = {k: v for k, v in gibbs.checkConvergence(MCMC).items() if k != "detail"} if MCMC.trace == "full" else None
            Post["Seconds"] = time.perf_counter() - t0
        finally:
            MCMC.close()
        Run[run] = Post
    return Run


def _stack(obj, par):
    true = np.asarray(obj["True"][par], dtype=np.float64)
    runs = [k for k in obj if k != "True"]
    esti = np.column_stack([np.asarray(obj[k][par], dtype=np.float64) for k in runs])
    if par in ("β", "beta") and esti.shape[0] == true.size + 1:
        esti = esti[1:]                                   # the intercept row is not part of the truth (src/SimTools.jl:501-506)
    return true, esti


def getMetrics(obj, *, par="a"):
    """src/SimTools.jl:500-523 over however many replications `obj` holds (the reference hard-codes 100)."""
    true, esti = _stack(obj, par)
    d = esti - true[:, None]
    return dict(Bias=float(np.mean(d)), Rmse=float(np.sqrt(np.mean(d ** 2))),
                Corr=float(np.mean([np.corrcoef(esti[:, k], true)[0, 1] for k in range(esti.shape[1])])))


def getMetrics2(obj, *, par="a"):
    """src/SimTools.jl:528-551"""
    true, esti = _stack(obj, par)
    d = esti - true[:, None]
    return dict(relativeBias=float(np.mean(d / true[:, None])), normalizedRmse=float(np.sqrt(np.mean(d ** 2)) / (esti.max() - esti.min())),
                Corr=float(np.mean([np.corrcoef(esti[:, k], true)[0, 1] for k in range(esti.shape[1])])))


def comparePara(Mcmc, *, par="a", digits=3):
    """src/SimTools.jl:389-413: the (Esti, True, |Diff|) table as an array (the reference prints it)."""
    true = _field(Mcmc.truePara, par).reshape(-1, order="F")
    esti = _field(Mcmc.Post.mean, par).reshape(-1, order="F")
    if par in ("β", "beta") and esti.size != true.size:
        nf = Mcmc.Cond.nFeat
        esti = esti.reshape(nf + 1, -1, order="F")[1:].reshape(-1, order="F")
    return np.round(np.column_stack([esti, true, np.abs(esti - true)]), digits)
