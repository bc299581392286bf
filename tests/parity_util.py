"""Test helpers: ctypes binding of the CPU oracle (oracle/liberm_oracle.so) and a runner that pushes the same seeded
inputs through the HIP library (via its C ABI) and through the oracle.  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

MODELS = {"mlirt": 0, "rtirt": 1, "crossqr": 2, "latentqr": 3, "null": 4, "cross": 5, "latent": 6}
BASE = {"null": "rtirt", "cross": "crossqr", "latent": "latentqr"}      # data generator / layout family of each variant
_DP = C.POINTER(C.c_double)


class orc_config(C.Structure):
    _fields_ = [("model", C.c_int32), ("nItem", C.c_int32), ("nSubj", C.c_int64), ("nFeat", C.c_int32), ("intercept", C.c_int32),
                ("onepl", C.c_int32), ("cov2one", C.c_int32), ("chain", C.c_int32), ("sigp_mode", C.c_int32), ("qRt", C.c_double),
                ("seed", C.c_uint64)]


class orc_data(C.Structure):
    _fields_ = [("Y", C.c_void_p), ("logT", C.c_void_p), ("X", C.c_void_p)]


class orc_state(C.Structure):
    _fields_ = [(n, _DP) for n in ("theta", "a", "b", "zeta", "lambda_", "sig2t", "beta", "Sigp", "rho", "nu", "omega")]


_orc = None


def oracle():
    global _orc
    if _orc is None:
        ge.build_oracle()
        lib = C.CDLL(ge.ORACLE_LIB)
        lib.orc_run.argtypes = [C.POINTER(orc_config), C.POINTER(orc_data), C.POINTER(orc_state), C.c_int64, C.c_int64,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_qr_width.argtypes = [C.POINTER(orc_config), C.c_int]
        lib.orc_loglik.argtypes = [C.POINTER(orc_config), C.POINTER(orc_data), C.POINTER(orc_state)]
        lib.orc_loglik.restype = C.c_double
        lib.orc_sample_batch.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_uint32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_sample_batch.restype = None
        lib.orc_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_philox.restype = None
        lib.orc_moments.argtypes = [C.POINTER(orc_config), C.POINTER(orc_data), C.POINTER(orc_state), C.c_int, C.c_void_p, C.c_void_p]
        lib.orc_moments.restype = None
        lib.orc_step.argtypes = [C.POINTER(orc_config), C.POINTER(orc_data), C.POINTER(orc_state), C.c_int, C.c_uint32]
        lib.orc_step.restype = None
        lib.orc_simulate_data.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 12
        lib.orc_simulate_data.restype = None
        lib.orc_set_threads.argtypes = [C.c_int]
        lib.orc_set_threads.restype = None
        _orc = lib
    return _orc


def orc_sample(which, n, par0=None, par1=None, *, seed=1234, site=15, sweep=1):
    out = np.empty(n, dtype=np.float64)
    p0 = None if par0 is None else np.ascontiguousarray(par0, dtype=np.float64)
    p1 = None if par1 is None else np.ascontiguousarray(par1, dtype=np.float64)
    oracle().orc_sample_batch(which, seed, site, sweep, n, None if p0 is None else p0.ctypes.data,
                              None if p1 is None else p1.ctypes.data, out.ctypes.data)
    return out


def orc_simulate(gen, N, J, F, *, a, b, lam=None, sig2t=None, rho=None, Sigp=None, beta=None, seed=4321, noise=0):
    """The oracle's restatement of setData* (gen 0 MlIrt, 1 RtIrt, 2 Null, 3 Cross, 4 Latent) with the device's stream addressing.
    Returns dict(X, theta, zeta, Y, logT), column-major."""
    def p(v):
        return None if v is None else np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1, order="F"))
    a_, b_, l_, s_, r_, S_, B_ = p(a), p(b), p(lam), p(sig2t), p(rho), p(Sigp), p(beta)
    X = np.zeros((N, max(F, 1)), order="F")
    th, ze = np.zeros(N), np.zeros(N)
    Y = np.zeros((N, J), dtype=np.uint8, order="F")
    logT = np.zeros((N, J), order="F")
    ptr = lambda v: None if v is None else v.ctypes.data
    oracle().orc_simulate_data(gen, noise, seed, N, J, F, ptr(a_), ptr(b_), ptr(l_), ptr(s_), ptr(r_), ptr(S_), ptr(B_),
                               X.ctypes.data, th.ctypes.data, ze.ctypes.data, Y.ctypes.data, logT.ctypes.data)
    return dict(X=X[:, :F], theta=th, zeta=ze, Y=Y, logT=logT)


class OracleProblem:
    """Holds data + state arrays (column-major, fp64) in the oracle's structs."""

    def __init__(self, model, Y, logT, X, state, *, qRt=0.5, intercept=False, onepl=False, cov2one=True, seed=1234, chain=0, sigp_mode=0):
        self.model = MODELS[model] if isinstance(model, str) else model
        self.N, self.J = Y.shape
        self.F = 0 if X is None else X.shape[1]
        self.Y = np.asfortranarray(Y.astype(np.uint8))
        self.logT = None if logT is None else np.asfortranarray(logT, dtype=np.float64)
        self.X = None if X is None else np.asfortranarray(X, dtype=np.float64)
        self.cfg = orc_config(self.model, self.J, self.N, self.F, int(intercept), int(onepl), int(cov2one), chain, int(sigp_mode), qRt, seed)
        self.data = orc_data(self.Y.ctypes.data, None if self.logT is None else self.logT.ctypes.data,
                             None if self.X is None else self.X.ctypes.data)
        N, J, F = self.N, self.J, self.F
        nb = {0: F + 1, 1: 2 * (F + 1), 2: 1, 3: F + 2, 4: 2 * (F + 1), 5: 1, 6: F + 2}[self.model]
        nnu = {2: N * J, 3: N}.get(self.model, 1)
        self.arr = dict(theta=np.zeros(N), a=np.ones(J), b=np.zeros(J), zeta=np.zeros(N), lambda_=np.zeros(J), sig2t=np.ones(J),
                        beta=np.zeros(nb), Sigp=np.array([1.0, 0, 0, 1.0]), rho=np.zeros(J), nu=np.ones(nnu), omega=np.zeros(N * J))
        for k, v in state.items():
            k = {"lam": "lambda_", "sigp": "Sigp"}.get(k, k)
            if v is not None:
                self.arr[k][:] = np.asarray(v, dtype=np.float64).reshape(-1, order="F")
        self.st = orc_state(*[self.arr[n].ctypes.data_as(_DP) for n, _ in orc_state._fields_])
        self.sweeps = 0

    def qr_width(self, with_nu):
        return oracle().orc_qr_width(C.byref(self.cfg), int(with_nu))

    def run(self, nsweeps, with_nu=False):
        N, J = self.N, self.J
        ra = np.zeros((nsweeps, N + 2 * J))
        rt = np.zeros((nsweeps, N + 2 * J))
        qr = np.zeros((nsweeps, self.qr_width(with_nu)))
        ll = np.zeros(nsweeps)
        oracle().orc_run(C.byref(self.cfg), C.byref(self.data), C.byref(self.st), self.sweeps, nsweeps,
                         ra.ctypes.data, rt.ctypes.data, qr.ctypes.data, int(with_nu), ll.ctypes.data)
        self.sweeps += nsweeps
        return dict(ra=ra, rt=rt, qr=qr, ll=ll)

    def moments(self, which, n1, n2=1):
        o1, o2 = np.zeros(n1), np.zeros(max(n2, 1))
        oracle().orc_moments(C.byref(self.cfg), C.byref(self.data), C.byref(self.st), which, o1.ctypes.data, o2.ctypes.data)
        return o1, o2

    def step(self, step, t):
        oracle().orc_step(C.byref(self.cfg), C.byref(self.data), C.byref(self.st), step, t)

    def loglik(self):
        return oracle().orc_loglik(C.byref(self.cfg), C.byref(self.data), C.byref(self.st))


# ---------------------------------------------------------------------------------------------------------------
def make_problem(model, N, J, F=3, seed=7, qRt=0.85):
    """Synthetic inputs in the style of setData* (src/SimTools.jl) plus a constructor-style initial state."""
    pkg = ge.load_package()
    Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=F, nIter=10, nChain=1, qRt=qRt)
    g = np.random.default_rng(seed)
    if model == "mlirt":
        tp = pkg.setTrueParaMlIrt(Cond, seed=g)
        D = pkg.setDataMlIrt(Cond, tp, seed=g)
        init = dict(theta=g.standard_normal(N), beta=g.standard_normal(F + 1))
        return D.Y, None, D.X, init, tp
    if model in ("rtirt", "null"):
        tp = pkg.setTrueParaRtIrt(Cond, seed=g)
        if model == "null":
            tp.beta = np.zeros((F, 2))          # the Null model has no covariate effects: theta, zeta ~ N(0, Sigp)
        D = pkg.setDataRtIrt(Cond, tp, seed=g)
        init = dict(theta=g.standard_normal(N), zeta=g.standard_normal(N), beta=g.standard_normal((F + 1, 2)), sigp=np.eye(2))
        if model == "null":
            init["beta"] = np.zeros((F + 1, 2))
        return D.Y, D.logT, D.X, init, tp
    if model in ("crossqr", "cross"):
        tp = pkg.setTrueParaRtIrtCross(Cond, seed=g)
        D = pkg.setDataRtIrtCross(Cond, tp, seed=g)
        init = dict(theta=g.standard_normal(N), zeta=g.standard_normal(N), rho=g.standard_normal(J), sigp=np.eye(2))
        return D.Y, D.logT, None, init, tp
    if model in ("latentqr", "latent"):
        tp = pkg.setTrueParaRtIrtLatent(Cond, seed=g)
        D = pkg.setDataRtIrtLatent(Cond, tp, seed=g)
        init = dict(theta=g.standard_normal(N), zeta=g.standard_normal(N), beta=g.standard_normal(F + 2), sigp=np.eye(2))
        return D.Y, D.logT, D.X, init, tp
    raise ValueError(model)


def run_device(model, Y, logT, X, init, nsweeps, *, precision="f64", qRt=0.85, seed=1234, intercept=False, onepl=False,
               cov2one=None, trace_full=True, n_chain=1, n_burnin=None, **opts):
    pkg = ge.load_package()
    L = pkg._lib
    N, J = Y.shape
    F = 0 if X is None else X.shape[1]
    if cov2one is None:
        cov2one = model not in ("latentqr", "latent")
    nb = nsweeps // 2 if n_burnin is None else n_burnin
    eng = L.Engine(model=MODELS[model], n_item=J, n_subj=N, n_feat=F, n_iter=nsweeps // n_chain, n_chain=n_chain, n_burnin=nb,
                   intercept=int(intercept), one_pl=int(onepl), cov2one=int(cov2one), q_rt=qRt, seed=seed,
                   precision={"f32": 0, "f64": 1}[precision], trace_mode=1 if trace_full else 0, **opts)
    eng.set_data(Y, logT, X)
    st = {("lambda_" if k == "lam" else k): v for k, v in init.items()}
    eng.set_state(**st)
    eng.run(nsweeps)
    out = dict(item=eng.item_trace(), ll=eng.trace(L.TRACE_LOGLIKE), state=eng.get_state(), engine=eng)
    if trace_full:
        out["ra"] = eng.trace(L.TRACE_RA)
        if model != "mlirt":
            out["rt"] = eng.trace(L.TRACE_RT)
        out["qr"] = eng.trace(L.TRACE_QR)
    return out


def run_pair(model, N, J, nsweeps, *, F=3, precision="f64", seed=7, qRt=0.85, **kw):
    """Same inputs through device and oracle.  Returns dict with device traces (rows x width) and oracle traces."""
    Y, logT, X, init, tp = make_problem(model, N, J, F, seed=seed, qRt=qRt)
    cov2one = kw.get("cov2one")
    if cov2one is None:
        cov2one = model not in ("latentqr", "latent")
    dev = run_device(model, Y, logT, X, init, nsweeps, precision=precision, qRt=qRt, **kw)
    op = OracleProblem(model, Y, logT, X, init, qRt=qRt, intercept=kw.get("intercept", False), onepl=kw.get("onepl", False),
                       cov2one=cov2one, seed=kw.get("seed", 1234), sigp_mode=kw.get("sigp_mode", 0))
    orc = op.run(nsweeps, with_nu=(model in ("latentqr", "crossqr")))
    d = dict(orc=orc, dev=dev, model=model)
    # device traces in Julia layout (nIter, width, nChain=1) -> rows x width
    d["dev_ra"] = dev["ra"][:, :, 0]
    if model != "mlirt":
        d["dev_rt"] = dev["rt"][:, :, 0]
    d["dev_qr"] = dev["qr"][:, :, 0]
    d["dev_ll"] = dev["ll"][:, 0, 0]
    return d


def rel_err(x, y, floor=1e-6):
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    return np.abs(x - y) / np.maximum(np.abs(y), floor)


def max_rel_err(res, floor=1e-6):
    e = [rel_err(res["dev_ra"], res["orc"]["ra"], floor).max()]
    if res["model"] != "mlirt":
        e.append(rel_err(res["dev_rt"], res["orc"]["rt"], floor).max())
    e.append(rel_err(res["dev_qr"], res["orc"]["qr"], floor).max())
    e.append(rel_err(res["dev_ll"], res["orc"]["ll"], floor).max())
    return float(max(e))
