"""Sampler objects and `sample!` for the MI355X engine -- host-side mirror of
/root/reference/src/GibbsRtIrt.pl.jl:35-472, src/GibbsRtIrtCross.pl.jl:55-353, src/GibbsRtIrtLatent.pl.jl:50-365.

Same struct names, fields (Cond, Data, truePara, Para, Post), constructor behaviour (always (re)initialises Para and
allocates Post), kwargs and error text as the reference.  `sample!` is spelled `sample_b` (PyJulia's convention for `!`)
and is also exported as `sample`.  All sampling happens in libertirt.so on the GPU; there is no CPU path here.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .base import InputPara, OutputDic, SimConditions

_PREC = {"f32": _lib.PREC_F32, "f64": _lib.PREC_F64}
_TRACE = {"summary": _lib.TRACE_SUMMARY, "full": _lib.TRACE_FULL}


class _OutputPost:
    """OutputPostMlIrt / OutputPost / OutputPostCrossQr / OutputPostRtIrtLatentQr
    (src/GibbsRtIrt.pl.jl:35-71, src/GibbsRtIrtCross.pl.jl:55-69, src/GibbsRtIrtLatent.pl.jl:50-64).
    ra, rt, qr: (nIter, width, nChain); logLike: (nIter, 1, nChain); mean: InputPara with flat vectors."""

    def __init__(self):
        self.ra = np.zeros(0)
        self.rt = np.zeros(0)
        self.qr = np.zeros(0)
        self.logLike = np.zeros(0)
        self.mean = InputPara()


class _GibbsBase:
    _model = None
    _cov2one_default = True
    _has_intercept = True

    def __init__(self, Cond: SimConditions, *, Data=None, truePara=None, Para=None, Post=None,
                 seed=1234, device=0, precision="f64", trace="full", chain_id=0, shard=None, **engine_opts):
        self.Cond = Cond
        self.Data = Data
        self.truePara = truePara
        self.seed = int(seed)
        self.device = int(device)
        self.precision = precision
        self.trace = trace
        self.chain_id = int(chain_id)
        if not 0 <= self.chain_id <= 255:
            raise ValueError("chain_id must be in 0 .. 255 (the random streams carry eight bits of it)")
        # shard = (rank, count, nSubjTotal, rowBase, transport): this process holds subjects [rowBase, rowBase + Cond.nSubj) of ONE chain
        # spread over `count` devices (include/ertirt.h erm_set_shard*); transport = the 128-byte id of _lib.rccl_unique_id() (the
        # library's in-stream RCCL all-gather) or a callable exchange(send_ptr, recv_ptr, nbytes) such as parallel.TorchExchange.
        # Cond.nSubj, Data, Para.theta / zeta / nu and the subject blocks of Post then describe the local subjects only.
        self.shard = shard
        self.engine_opts = dict(engine_opts)
        self._engine = None
        self._engine_key = None
        self._data_on_device = False
        self.farm = None
        self.Para = None
        self.setInitialValues()          # constructors always overwrite Para (src/GibbsRtIrt.pl.jl:100-102)
        if shard is not None:
            self._shard_initial_values()
        self.Post = _OutputPost()

    def _shard_initial_values(self):
        """Initial values of a shard = the UNSHARDED sampler's initial values restricted to the local subjects: the whole-data-set
        state is generated (same seed on every rank, so item and structural entries agree bit for bit) and theta / zeta are cut."""
        import dataclasses
        rank, count, ntot, base, _ = self.shard
        local = self.Cond
        if not (0 <= base and base + local.nSubj <= ntot):
            raise ValueError("shard: local subjects must lie inside [0, nSubjTotal)")
        self.Cond = dataclasses.replace(local, nSubj=int(ntot))
        try:
            self.setInitialValues()
        finally:
            self.Cond = local
        for f in ("theta", "zeta"):
            v = getattr(self.Para, f)
            if np.size(v) == ntot:
                setattr(self.Para, f, np.ascontiguousarray(np.asarray(v)[base:base + local.nSubj]))

    # -- per-model hooks
    def setInitialValues(self):
        raise NotImplementedError

    def _rng(self):
        return np.random.default_rng(np.random.SeedSequence([self.seed, self.chain_id, 0x1217]))

    def _state_for_engine(self):
        P = self.Para
        d = dict(theta=P.theta, a=P.a, b=P.b)
        if self._model != _lib.MODEL_MLIRT:
            d.update(zeta=P.zeta, lambda_=P.lam, sig2t=P.sig2t, sigp=np.asarray(P.Sigp, dtype=np.float64).reshape(-1, order="F"))
        if P.beta.size:
            d["beta"] = np.asarray(P.beta, dtype=np.float64).reshape(-1, order="F")
        if P.rho.size:
            d["rho"] = P.rho
        if P.nu.size and self._model in (_lib.MODEL_CROSSQR, _lib.MODEL_LATENTQR):
            d["nu"] = np.asarray(P.nu, dtype=np.float64).reshape(-1, order="F")
        return d

    def _engine_for(self, intercept, onepl, cov2one, *, upload=True):
        key = (bool(intercept), bool(onepl), bool(cov2one))
        if self._engine is not None and self._engine_key == key:
            return self._engine
        resident = None
        if self._engine is not None:
            if not upload or self._data_on_device:
                resident = self._engine.get_data()     # the data set lives on the device only: carry it over to the new engine
            self._engine.close()
        C = self.Cond
        if self.Data is None and upload and resident is None:
            raise ValueError("Data is required")
        eng = _lib.Engine(model=self._model, n_item=C.nItem, n_subj=C.nSubj, n_feat=C.nFeat, n_iter=C.nIter, n_chain=C.nChain,
                          n_burnin=C.nBurnin, intercept=int(intercept), one_pl=int(onepl), cov2one=int(cov2one), q_rt=C.qRt,
                          seed=self.seed, chain_id=self.chain_id, device=self.device, precision=_PREC[self.precision],
                          trace_mode=_TRACE[self.trace], **self.engine_opts)
        self._engine, self._engine_key = eng, key
        if self.shard is not None:
            rank, count, ntot, base, transport = self.shard
            if callable(transport):
                eng.set_shard(rank, count, ntot, base, transport)
            else:
                eng.set_shard_rccl(rank, count, ntot, base, transport)
        if resident is not None:
            eng.set_data(*resident)
            return eng
        if not upload:
            return eng
        D = self.Data
        Y = np.asarray(D.Y)
        if Y.shape != (C.nSubj, C.nItem):
            raise ValueError(f"Data.Y must be {C.nSubj}x{C.nItem}, got {Y.shape}")
        logT = None
        if self._model != _lib.MODEL_MLIRT:
            logT = np.asarray(D.logT, dtype=np.float64)
            if logT.shape != (C.nSubj, C.nItem):
                raise ValueError(f"Data.logT must be {C.nSubj}x{C.nItem}, got {logT.shape}")
        X = None
        if self._model not in (_lib.MODEL_CROSSQR, _lib.MODEL_CROSS, _lib.MODEL_NULL) and C.nFeat > 0:
            X = np.asarray(D.X, dtype=np.float64)
            if X.shape != (C.nSubj, C.nFeat):
                raise ValueError(f"Data.X must be {C.nSubj}x{C.nFeat}, got {X.shape}")
        eng.set_data(Y, logT, X)
        self._engine, self._engine_key = eng, key
        return eng

    def _fill_post(self, eng):
        C = self.Cond
        Post = self.Post
        full = self.trace == "full"
        Post.logLike = eng.trace(_lib.TRACE_LOGLIKE)
        if full:
            Post.ra = eng.trace(_lib.TRACE_RA)
            if self._model != _lib.MODEL_MLIRT:
                Post.rt = eng.trace(_lib.TRACE_RT)
            try:
                Post.qr = eng.trace(_lib.TRACE_QR)
            except _lib.ErmError:
                if self._model != _lib.MODEL_CROSSQR:
                    raise
                # vec(nu) per sweep did not fit the device budget: Post.qr keeps [rho; vec(Sigp)] (the reference's first nItem+4 columns)
                it = eng.item_trace()
                Post.qr = np.asfortranarray(it[:, 4 * C.nItem:].reshape(C.nIter, C.nChain, -1).transpose(0, 2, 1))
        Post.item_trace = eng.item_trace()
        m = eng.get_mean()
        mean = InputPara(theta=m["theta"], a=m["a"], b=m["b"])
        if self._model != _lib.MODEL_MLIRT:
            mean.zeta, mean.lam, mean.sig2t = m["zeta"], m["lambda_"], m["sig2t"]
            mean.Sigp = m["sigp"]
        if m["beta"] is not None:
            mean.beta = m["beta"]
        if self._model in (_lib.MODEL_CROSSQR, _lib.MODEL_CROSS):
            mean.rho = m["rho"]
        if m["nu"] is not None:
            mean.nu = m["nu"]
        Post.mean = mean

    def _update_para(self, eng):
        s = eng.get_state()
        C = self.Cond
        P = self.Para
        P.theta, P.a, P.b = s["theta"], s["a"], s["b"]
        if self._model != _lib.MODEL_MLIRT:
            P.zeta, P.lam, P.sig2t = s["zeta"], s["lambda_"], s["sig2t"]
            P.Sigp = s["sigp"].reshape(2, 2, order="F")
        if s["beta"] is not None:
            P.beta = s["beta"].reshape(C.nFeat + 1, 2, order="F") if self._model in (_lib.MODEL_RTIRT, _lib.MODEL_NULL) else s["beta"]
        if self._model in (_lib.MODEL_CROSSQR, _lib.MODEL_CROSS):
            P.rho = s["rho"]
        if self._model == _lib.MODEL_CROSSQR:
            P.nu = s["nu"].reshape(C.nSubj, C.nItem, order="F")
        if self._model == _lib.MODEL_LATENTQR:
            P.nu = s["nu"]

    def timing(self):
        return self._engine.timing() if self._engine is not None else None

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        if getattr(self, "farm", None) is not None:
            self.farm.close()
            self.farm = None


def simulateData(MCMC: _GibbsBase, truePara: InputPara, *, type="norm", seed=4321, pull=True, pull_truth=True, intercept=False, itemtype="2pl", cov2one=None):
    """setData* on the device (erm_simulate_data): generates X, theta, zeta, Y, logT from `truePara` straight into the engine's
    resident buffers -- no host generation, no upload.  truePara.theta / .zeta receive the generated truth (pull_truth=False leaves it on the
    device); with pull=True the data set is also copied to MCMC.Data (the host-side getLogLikelihood / getDicHost need it; getDic does not)."""
    from .base import InputData
    # (the engine is keyed by the sample! kwargs: give the ones the following sample! will use, or the data set is carried over through the host)
    eng = MCMC._engine_for(intercept, itemtype == "1pl", MCMC._cov2one_default if cov2one is None else cov2one, upload=False)
    noise = {"norm": 0, "tail": 1, "skew": 2}[type]
    truth = dict(a=truePara.a, b=truePara.b)
    if MCMC._model != _lib.MODEL_MLIRT:
        truth.update(lambda_=truePara.lam, sig2t=truePara.sig2t if truePara.sig2t.size else np.ones(MCMC.Cond.nItem))
        if np.size(truePara.Sigp):
            truth["sigp"] = np.asarray(truePara.Sigp, dtype=np.float64).reshape(-1, order="F")
    if truePara.beta.size:
        truth["beta"] = np.asarray(truePara.beta, dtype=np.float64).reshape(-1, order="F")
    if truePara.rho.size:
        truth["rho"] = truePara.rho
    tz = eng.simulate_data(seed=seed, noise=noise, pull_truth=pull_truth, **truth)
    if pull_truth:
        truePara.theta, truePara.zeta = tz
    MCMC.truePara = truePara
    MCMC._data_on_device = True
    if pull:
        Y, logT, X = eng.get_data()
        MCMC.Data = InputData(Y=Y, T=np.exp(logT) if logT is not None else (), X=X if X is not None else ())
    return MCMC


def _sample_farm(MCMC: _GibbsBase, intercept, onepl, cov2one, devices):
    """sample! with Cond.nChain INDEPENDENT chains, chain l on GPU devices[l mod len(devices)] (erm_farm_*, include/ertirt.h): the chains
    run concurrently inside the library, Post.ra / rt / qr / logLike get chain l in slab l, and Post.mean is the joint mean over
    iterations and chains (src/GibbsRtIrt.pl.jl:327-343) reduced over the devices by one RCCL all-reduce.  Chain 0 starts from
    MCMC.Para (the constructor's setInitialValues); chain l > 0 from its own setInitialValues draw on random stream l."""
    import copy
    C = MCMC.Cond
    if MCMC.shard is not None:
        raise ValueError("a subject-sharded sampler cannot also farm chains")
    if MCMC.Data is None:
        raise ValueError("Data is required")
    devs = [int(devices[l % len(devices)]) for l in range(C.nChain)]
    farm = _lib.Farm(devs, model=MCMC._model, n_item=C.nItem, n_subj=C.nSubj, n_feat=C.nFeat, n_iter=C.nIter, n_chain=1, n_burnin=C.nBurnin,
                     intercept=int(intercept), one_pl=int(onepl), cov2one=int(cov2one), q_rt=C.qRt, seed=MCMC.seed, precision=_PREC[MCMC.precision],
                     trace_mode=_TRACE[MCMC.trace], **MCMC.engine_opts)
    D = MCMC.Data
    logT = None if MCMC._model == _lib.MODEL_MLIRT else np.asarray(D.logT, dtype=np.float64)
    X = None
    if MCMC._model not in (_lib.MODEL_CROSSQR, _lib.MODEL_CROSS, _lib.MODEL_NULL) and C.nFeat > 0:
        X = np.asarray(D.X, dtype=np.float64)
    farm.set_data(np.asarray(D.Y), logT, X)
    for l in range(C.nChain):
        if l == 0:
            farm.set_state(0, **MCMC._state_for_engine())
        else:
            other = copy.copy(MCMC)
            other.chain_id = l
            other.setInitialValues()
            farm.set_state(l, **other._state_for_engine())
    farm.run(C.nIter)
    MCMC._fill_post(farm)
    MCMC._update_para(farm.engine(0))
    MCMC.farm = farm
    return MCMC


def sample_b(MCMC: _GibbsBase, *, intercept=False, itemtype="2pl", cov2one=None, devices=None, fill=True):
    """sample!(MCMC; intercept, itemtype, cov2one) -- src/GibbsRtIrt.pl.jl:210,278; Cross :265; Latent :271.
    Runs Cond.nIter * Cond.nChain sweeps (the reference's interleaved `for m in 1:nIter, l in 1:nChain` loop over ONE
    shared Para), fills MCMC.Post, leaves the final state in MCMC.Para and returns MCMC.
    devices = [gpu ordinals]: the nChain chains become INDEPENDENT chains farmed over those GPUs instead (see _sample_farm).
    fill = False leaves Post and Para untouched: traces, running means and the final state stay on the device, where getDic,
    checkConvergence and MCMC._engine.get_mean(which) read them (runSimulation's replications cross the boundary with summaries only)."""
    if itemtype not in ("1pl", "2pl"):
        raise ValueError("Invalid input: the item type must be '1pl' or '2pl'.")   # same text as :213,281
    if cov2one is None:
        cov2one = MCMC._cov2one_default
    if intercept and not MCMC._has_intercept:
        raise TypeError(f"sample! for {type(MCMC).__name__} has no `intercept` keyword")
    if devices is not None:
        return _sample_farm(MCMC, intercept, itemtype == "1pl", cov2one, list(devices))
    if MCMC.farm is not None:
        MCMC.farm.close()
        MCMC.farm = None
    eng = MCMC._engine_for(intercept, itemtype == "1pl", cov2one)
    eng.reset_trace()
    eng.set_state(**MCMC._state_for_engine())
    eng.run(MCMC.Cond.nIter * MCMC.Cond.nChain)
    if fill:
        MCMC._fill_post(eng)
        MCMC._update_para(eng)
    return MCMC


sample = sample_b


class GibbsMlIrt(_GibbsBase):
    """src/GibbsRtIrt.pl.jl:76-106.  theta's prior variance is 1 (the reference never sets Para.Σp for this model:
    :85-91, :228; its likelihood uses Normal(mu, 1.) :201)."""
    _model = _lib.MODEL_MLIRT

    def setInitialValues(self):
        C, g = self.Cond, self._rng()
        self.Para = InputPara(theta=g.standard_normal(C.nSubj), a=np.ones(C.nItem), b=np.zeros(C.nItem),
                              beta=g.standard_normal(C.nFeat + 1))
        return self


class GibbsRtIrt(_GibbsBase):
    """src/GibbsRtIrt.pl.jl:114-146"""
    _model = _lib.MODEL_RTIRT

    def setInitialValues(self):
        C, g = self.Cond, self._rng()
        self.Para = InputPara(theta=g.standard_normal(C.nSubj), a=np.ones(C.nItem), b=np.zeros(C.nItem),
                              zeta=g.standard_normal(C.nSubj), lam=np.zeros(C.nItem), sig2t=np.ones(C.nItem),
                              beta=g.standard_normal((C.nFeat + 1, 2)), Sigp=np.eye(2))
        return self


class GibbsRtIrtCrossQr(_GibbsBase):
    """src/GibbsRtIrtCross.pl.jl:115-147"""
    _model = _lib.MODEL_CROSSQR
    _has_intercept = False

    def setInitialValues(self):
        C, g = self.Cond, self._rng()
        self.Para = InputPara(theta=g.standard_normal(C.nSubj), a=np.ones(C.nItem), b=np.zeros(C.nItem),
                              zeta=g.standard_normal(C.nSubj), lam=np.zeros(C.nItem), sig2t=np.ones(C.nItem),
                              rho=g.standard_normal(C.nItem), Sigp=np.eye(2))
        return self


class GibbsRtIrtLatentQr(_GibbsBase):
    """src/GibbsRtIrtLatent.pl.jl:105-137 (sample! default cov2one = false, :271)"""
    _model = _lib.MODEL_LATENTQR
    _cov2one_default = False

    def setInitialValues(self):
        C, g = self.Cond, self._rng()
        self.Para = InputPara(theta=g.standard_normal(C.nSubj), a=np.ones(C.nItem), b=np.zeros(C.nItem),
                              zeta=g.standard_normal(C.nSubj), lam=np.zeros(C.nItem), sig2t=np.ones(C.nItem),
                              beta=g.standard_normal(C.nFeat + 2), Sigp=np.eye(2))
        return self


class GibbsRtIrtNull(_GibbsBase):
    """src/GibbsRtIrt.pl.jl:151-183: no latent regression (beta = 0 every sweep, :380), theta ~ N(0, Sigp11), zeta ~ N(0, 1)."""
    _model = _lib.MODEL_NULL
    _has_intercept = False

    def setInitialValues(self):
        C, g = self.Cond, self._rng()
        self.Para = InputPara(theta=g.standard_normal(C.nSubj), a=np.ones(C.nItem), b=np.zeros(C.nItem),
                              zeta=g.standard_normal(C.nSubj), lam=np.zeros(C.nItem), sig2t=np.ones(C.nItem), Sigp=np.eye(2))
        return self


class GibbsRtIrtCross(_GibbsBase):
    """src/GibbsRtIrtCross.pl.jl:77-110: cross-relation rho without quantile weights."""
    _model = _lib.MODEL_CROSS
    _has_intercept = False

    def setInitialValues(self):
        C, g = self.Cond, self._rng()
        self.Para = InputPara(theta=g.standard_normal(C.nSubj), a=np.ones(C.nItem), b=np.zeros(C.nItem),
                              zeta=g.standard_normal(C.nSubj), lam=np.zeros(C.nItem), sig2t=np.ones(C.nItem),
                              rho=g.standard_normal(C.nItem), Sigp=np.eye(2))
        return self


class GibbsRtIrtLatent(_GibbsBase):
    """src/GibbsRtIrtLatent.pl.jl:70-102 (sample! default cov2one = false, :168): zeta regressed on [1 X theta], beta drawn."""
    _model = _lib.MODEL_LATENT
    _cov2one_default = False

    def setInitialValues(self):
        C, g = self.Cond, self._rng()
        self.Para = InputPara(theta=g.standard_normal(C.nSubj), a=np.ones(C.nItem), b=np.zeros(C.nItem),
                              zeta=g.standard_normal(C.nSubj), lam=np.zeros(C.nItem), sig2t=np.ones(C.nItem),
                              beta=g.standard_normal(C.nFeat + 2), Sigp=np.eye(2))
        return self


# README.md:22,95 names `GibbsRtIrtQuantile`; the export is commented out in the reference (src/ExtendedRtIrtModeling.jl:65) and
# the only live type with that API (X, beta, Sigp, qRt) is GibbsRtIrtLatentQr.
GibbsRtIrtQuantile = GibbsRtIrtLatentQr


# ------------------------------------------------------------------------------------------------------------------
# log-likelihoods at a parameter point and DIC (host-side post-processing, numpy fp64)
# ------------------------------------------------------------------------------------------------------------------
def _log1pexp(x):
    return np.where(x > 0, x + np.log1p(np.exp(-np.abs(x))), np.log1p(np.exp(-np.abs(x))))


def _norm_logpdf(x, mu, sd):
    return -0.5 * np.log(2 * np.pi) - np.log(sd) - 0.5 * ((x - mu) / sd) ** 2


def getLogLikelihood(MCMC: _GibbsBase, P: InputPara) -> float:
    """getLogLikelihoodMlIrt / RtIrt / RtIrtNull / RtIrtCross(Qr) / RtIrtLatent(Qr) evaluated at P
    (src/GibbsRtIrt.pl.jl:195-204,262-272,351-362; src/GibbsRtIrtCross.pl.jl:158-170,240-258; src/GibbsRtIrtLatent.pl.jl:151-162,243-264)."""
    C, D = MCMC.Cond, MCMC.Data
    Y = np.asarray(D.Y, dtype=np.float64)
    th, a, b = P.theta, P.a, P.b
    pr = a[None, :] * (th[:, None] - b[None, :])
    ll = np.sum(Y * pr - _log1pexp(pr))
    m = MCMC._model
    if m == _lib.MODEL_MLIRT:
        x = np.column_stack([np.ones(C.nSubj), D.X])
        return float(ll + np.sum(_norm_logpdf(th, x @ P.beta, 1.0)))
    q = C.qRt
    k1, k2 = (1 - 2 * q) / (q * (1 - q)), 2 / (q * (1 - q))
    logT = np.asarray(D.logT, dtype=np.float64)
    nu_lat = P.nu
    if m in (_lib.MODEL_NULL, _lib.MODEL_CROSS, _lib.MODEL_LATENT):
        k1, k2, nu_lat = 0.0, 1.0, 1.0            # no quantile weights
    if m == _lib.MODEL_CROSS:
        mut = P.lam[None, :] - P.zeta[:, None] - th[:, None] * P.rho[None, :]
        ll += np.sum(_norm_logpdf(logT, mut, np.sqrt(P.sig2t)[None, :]))
    elif m == _lib.MODEL_CROSSQR:
        e = np.asarray(P.nu).reshape(C.nSubj, C.nItem, order="F")
        mut = P.lam[None, :] - P.zeta[:, None] - th[:, None] * P.rho[None, :] + k1 * e
        ll += np.sum(_norm_logpdf(logT, mut, np.sqrt(P.sig2t[None, :] * (k2 * e))))
    else:
        ll += np.sum(_norm_logpdf(logT, P.lam[None, :] - P.zeta[:, None], np.sqrt(P.sig2t)[None, :]))
    S = np.asarray(P.Sigp, dtype=np.float64).reshape(2, 2, order="F")
    if m in (_lib.MODEL_RTIRT, _lib.MODEL_CROSSQR, _lib.MODEL_NULL, _lib.MODEL_CROSS):
        eta = np.column_stack([th, P.zeta])
        if m == _lib.MODEL_RTIRT:
            x = np.column_stack([np.ones(C.nSubj), D.X])
            eta = eta - x @ np.asarray(P.beta).reshape(C.nFeat + 1, 2, order="F")
        Si = np.linalg.inv(S)
        quad = np.einsum("ia,ab,ib->i", eta, Si, eta)
        ll += np.sum(-np.log(2 * np.pi) - 0.5 * np.log(np.linalg.det(S)) - 0.5 * quad)
    else:
        x = np.column_stack([np.ones(C.nSubj), D.X, th])
        ll += np.sum(_norm_logpdf(P.zeta, x @ P.beta + k1 * nu_lat, np.sqrt(S[1, 1] * k2 * nu_lat)))
    return float(ll)


def getDicHost(MCMC: _GibbsBase) -> OutputDic:
    """getDic evaluated with numpy on the host from MCMC.Data and MCMC.Post (the test twin of the device path below; needs the data set and
    Post.mean on the host)."""
    Dhat = -2.0 * getLogLikelihood(MCMC, MCMC.Post.mean)
    Dbar = -2.0 * float(np.mean(MCMC.Post.logLike))
    pD = Dbar - Dhat
    return OutputDic(pD=pD, DIC=Dbar + pD)


def getDic(MCMC: _GibbsBase) -> OutputDic:
    """src/GibbsRtIrt.pl.jl:432-458 (and Cross :329-353, Latent :341-365): D̂ = -2 logLik(Post.mean),
    D̄ = -2 mean(Post.logLike) over ALL iterations (burn-in included, as the reference does).  Computed by the engine from device-resident
    state (erm_get_dic / erm_farm_get_dic: the logLike rows, the running sums behind Post.mean and ONE evaluation pass over the resident data
    set): neither the data set nor an N-wide mean has to be on the host.  Call it after sample!, before the sampler is closed."""
    src = getattr(MCMC, "farm", None) or MCMC._engine
    if src is None:
        raise ValueError("run sample! first (getDic reads the engine's resident state)")
    d = src.dic()
    return OutputDic(pD=d["pD"], DIC=d["DIC"])


def ess_rhat(x: np.ndarray):
    """Split-R-hat and effective sample size (Geyer's initial monotone sequence over the split chains; BDA3 sec. 11.4-11.5, the
    non-rank-normalised estimator) of draws x[(iteration, chain)] -- the host twin of the device kernel `diag_kernel`, used by its
    tests and for traces that are not resident on the device."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    T, C = x.shape
    n = T // 2
    seq = np.stack([x[:n, l] if h == 0 else x[T - n:, l] for l in range(C) for h in (0, 1)])      # (M, n)
    M = seq.shape[0]
    mu = seq.mean(axis=1)
    d = seq - mu[:, None]
    W = np.mean(np.sum(d * d, axis=1) / (n - 1))
    Bn = np.sum((mu - mu.mean()) ** 2) / (M - 1)
    varp = W * (n - 1) / n + Bn
    if not W > 0:
        return float("nan"), float("nan")

    def rho(t):
        return 1.0 - (W - np.mean(np.sum(d[:, :n - t] * d[:, t:], axis=1) / n)) / varp

    total, prev, t = 0.0, np.inf, 0
    while t + 1 < n:
        P = rho(t) + rho(t + 1)
        if not P > 0:
            break
        P = min(P, prev)
        prev = P
        total += P
        t += 2
    return M * n / (-1.0 + 2.0 * total), float(np.sqrt(varp / W))


def checkConvergence(MCMC: _GibbsBase, *, detail=True) -> dict:
    """src/SimTools.jl:419-443: share of the ra / rt / qr columns with ESS > 400 and R-hat < 1.1 after burn-in.  The reference runs
    MCMCChains' `ess_rhat` on the host; here both statistics come from the device-resident traces (erm_get_diagnostics; split-R-hat and
    Geyer's initial-monotone-sequence ESS, not rank-normalised -- MCMCChains' version is not pinned by the reference).  Columns that
    never move (NaN) are left out of the denominators, as the reference does for qr.  detail=False also COUNTS on the device
    (erm_get_convergence): eight integers cross the boundary instead of the N-wide ess / rhat vectors."""
    eng = MCMC._engine
    if eng is None:
        raise ValueError("run sample! first")
    ess_n = rhat_n = ess_ok = rhat_ok = 0
    out = {}
    for name, which in (("ra", _lib.TRACE_RA), ("rt", _lib.TRACE_RT), ("qr", _lib.TRACE_QR)):
        if which == _lib.TRACE_RT and MCMC._model == _lib.MODEL_MLIRT:
            continue
        try:
            if detail:
                ess, rhat = eng.diagnostics(which)
                c = (int(np.sum(~np.isnan(ess))), int(np.sum(ess > 400)), int(np.sum(~np.isnan(rhat))), int(np.sum(rhat < 1.1)))
                out[name] = (ess, rhat)
            else:
                c = eng.convergence(which)
        except _lib.ErmError:
            if which != _lib.TRACE_QR:
                raise
            continue                                    # CrossQr without a resident nu trace
        ess_n += c[0]; ess_ok += c[1]; rhat_n += c[2]; rhat_ok += c[3]
    res = dict(ess=100.0 * ess_ok / max(ess_n, 1), rhat=100.0 * rhat_ok / max(rhat_n, 1), essN=f"{ess_ok} / {ess_n}", rhatN=f"{rhat_ok} / {rhat_n}")
    if detail:
        res["detail"] = out
    return res


def coef(MCMC: _GibbsBase) -> dict:
    """Posterior-mean tables of `coef` (src/GibbsRtIrt.pl.jl:479-538) as plain arrays (pretty-printing is out of scope)."""
    C, M = MCMC.Cond, MCMC.Post.mean
    out = {"a": M.a, "b": M.b}
    if MCMC._model != _lib.MODEL_MLIRT:
        out.update({"λ": M.lam, "σ²t": M.sig2t, "Σp": np.asarray(M.Sigp).reshape(2, 2, order="F")})
    if MCMC._model in (_lib.MODEL_RTIRT, _lib.MODEL_NULL):
        out["β"] = np.asarray(M.beta).reshape(C.nFeat + 1, 2, order="F")
    elif M.beta.size:
        out["β"] = M.beta
    if MCMC._model in (_lib.MODEL_CROSSQR, _lib.MODEL_CROSS):
        out["ρ"] = M.rho
    return out


def precis(MCMC: _GibbsBase) -> dict:
    """Mean / sd / 2.5% / 97.5% of the item-level and structural traces after burn-in (`precis`, src/GibbsRtIrt.pl.jl:545-675,
    without the MCMCChains ESS/R-hat columns)."""
    C = MCMC.Cond
    it = MCMC.Post.item_trace[C.nBurnin * C.nChain:]
    J = C.nItem
    names = [f"a[{j+1}]" for j in range(J)] + [f"b[{j+1}]" for j in range(J)] + [f"λ[{j+1}]" for j in range(J)] + \
            [f"σ²t[{j+1}]" for j in range(J)] + [f"qr[{k+1}]" for k in range(it.shape[1] - 4 * J)]
    return {"names": names, "mean": it.mean(0), "std": it.std(0, ddof=1), "q025": np.quantile(it, 0.025, axis=0),
            "q975": np.quantile(it, 0.975, axis=0)}
