"""ctypes binding of libertirt.so (include/ertirt.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ERM_LIB_PATH") or os.path.join(_HERE, "libertirt.so")      # ERM_LIB_PATH: diagnostics (library build variants)

MODEL_MLIRT, MODEL_RTIRT, MODEL_CROSSQR, MODEL_LATENTQR = 0, 1, 2, 3
MODEL_NULL, MODEL_CROSS, MODEL_LATENT = 4, 5, 6          # the non-quantile variants
PREC_F32, PREC_F64 = 0, 1
TRACE_SUMMARY, TRACE_FULL = 0, 1
TRACE_RA, TRACE_RT, TRACE_QR, TRACE_LOGLIKE = 0, 1, 2, 3

EXPORTS = [
    "erm_create", "erm_destroy", "erm_set_data", "erm_set_state", "erm_get_state", "erm_run", "erm_rows_done",
    "erm_reset_trace", "erm_trace_width", "erm_get_trace", "erm_item_trace_width", "erm_get_item_trace", "erm_get_mean",
    "erm_post_count", "erm_get_diagnostics", "erm_simulate_data", "erm_get_truth", "erm_get_data", "erm_get_timing", "erm_last_error", "erm_version", "erm_debug_sample", "erm_sample_gig",
    "erm_set_shard", "erm_copy", "erm_rccl_unique_id", "erm_set_shard_rccl",
    "erm_farm_create", "erm_farm_destroy", "erm_farm_chains", "erm_farm_engine", "erm_farm_set_data", "erm_farm_set_state", "erm_farm_get_state",
    "erm_farm_run", "erm_farm_reset_trace", "erm_farm_get_trace", "erm_farm_get_mean", "erm_farm_post_count", "erm_farm_used_rccl", "erm_farm_get_timing",
    "erm_get_dic", "erm_set_seed", "erm_farm_get_dic", "erm_farm_set_seed", "erm_abi_version", "erm_debug_invwishart", "erm_get_convergence",
]
ABI_VERSION = 4            # ERM_ABI_VERSION of the include/ertirt.h these ctypes structs mirror


class ErmError(RuntimeError):
    pass


class erm_config(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("n_item", C.c_int32), ("n_subj", C.c_int64), ("n_feat", C.c_int32), ("n_iter", C.c_int32),
        ("n_chain", C.c_int32), ("n_burnin", C.c_int32), ("intercept", C.c_int32), ("one_pl", C.c_int32),
        ("cov2one", C.c_int32), ("sigp_mode", C.c_int32), ("chain_id", C.c_int32), ("q_rt", C.c_double),
        ("seed", C.c_uint64), ("device", C.c_int32), ("precision", C.c_int32), ("trace_mode", C.c_int32),
        ("lanes_per_row", C.c_int32), ("block_threads", C.c_int32), ("grid_blocks", C.c_int32), ("profile", C.c_int32),
        ("flags", C.c_int32), ("nu_trace_max_gb", C.c_double),
    ]


FLAG_NO_FUSE, FLAG_NO_GRAPH, FLAG_FARM_FORCE_RCCL, FLAG_NO_PERSIST, FLAG_TEST_PERSIST_TIMEOUT = 1, 2, 4, 8, 16


class erm_farm_timing(C.Structure):
    _fields_ = [("run_wall_ms", C.c_double), ("gather_ms", C.c_double), ("allreduce_ms", C.c_double), ("comm_init_ms", C.c_double), ("rccl_ranks", C.c_int32), ("n_devices", C.c_int32)]


_DP = C.POINTER(C.c_double)


class erm_state(C.Structure):
    _fields_ = [(n, _DP) for n in ("theta", "a", "b", "zeta", "lambda_", "sig2t", "beta", "sigp", "rho", "nu")]


class erm_timing(C.Structure):
    _fields_ = [
        ("run_ms", C.c_double), ("pass_ms_total", C.c_double), ("event_overhead_ms", C.c_double), ("pass_launches", C.c_int64), ("sweeps", C.c_int64),
        ("lanes_per_row", C.c_int32), ("block_threads", C.c_int32), ("grid_blocks", C.c_int32), ("lds_bytes", C.c_int32),
        ("cu_count", C.c_int32), ("persistent", C.c_int32), ("persist_fallbacks", C.c_int32), ("reserved_", C.c_int32),
    ]


# int (*erm_exchange_fn)(void* user, const void* dev_send, void* dev_recv, size_t bytes_per_rank)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

_lib = None


def load():
    """Load libertirt.so; raises ErmError if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ErmError(f"HIP extension missing: {LIB_PATH} (build it with `python -c 'import __graft_entry__ as g; g.build()'`)")
    lib = C.CDLL(LIB_PATH)
    lib.erm_abi_version.restype = C.c_int
    if lib.erm_abi_version() != ABI_VERSION:
        raise ErmError(f"{LIB_PATH} has struct layout version {lib.erm_abi_version()}, this binding was written against {ABI_VERSION}: rebuild the library")
    H = C.c_void_p
    lib.erm_create.argtypes = [C.POINTER(erm_config), C.POINTER(H)]
    lib.erm_destroy.argtypes = [H]
    lib.erm_destroy.restype = None
    lib.erm_set_data.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.erm_set_state.argtypes = [H, C.POINTER(erm_state)]
    lib.erm_get_state.argtypes = [H, C.POINTER(erm_state)]
    lib.erm_run.argtypes = [H, C.c_int64]
    lib.erm_rows_done.argtypes = [H]
    lib.erm_rows_done.restype = C.c_int64
    lib.erm_reset_trace.argtypes = [H]
    lib.erm_trace_width.argtypes = [H, C.c_int]
    lib.erm_trace_width.restype = C.c_int64
    lib.erm_get_trace.argtypes = [H, C.c_int, C.c_void_p]
    lib.erm_item_trace_width.argtypes = [H]
    lib.erm_item_trace_width.restype = C.c_int64
    lib.erm_get_item_trace.argtypes = [H, C.c_void_p]
    lib.erm_get_mean.argtypes = [H, C.POINTER(erm_state)]
    lib.erm_post_count.argtypes = [H]
    lib.erm_post_count.restype = C.c_int64
    lib.erm_get_diagnostics.argtypes = [H, C.c_int, C.c_void_p, C.c_void_p]
    lib.erm_simulate_data.argtypes = [H, C.POINTER(erm_state), C.c_uint64, C.c_int]
    lib.erm_get_truth.argtypes = [H, C.c_void_p, C.c_void_p]
    lib.erm_get_data.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.erm_get_timing.argtypes = [H, C.POINTER(erm_timing)]
    lib.erm_last_error.restype = C.c_char_p
    lib.erm_version.restype = C.c_char_p
    lib.erm_debug_sample.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
    lib.erm_sample_gig.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_void_p]
    lib.erm_set_shard.argtypes = [H, C.c_int, C.c_int, C.c_int64, C.c_int64, EXCHANGE_FN, C.c_void_p]
    lib.erm_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.erm_rccl_unique_id.argtypes = [C.c_void_p]
    lib.erm_set_shard_rccl.argtypes = [H, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
    lib.erm_farm_create.argtypes = [C.POINTER(erm_config), C.POINTER(C.c_int32), C.c_int32, C.POINTER(H)]
    lib.erm_farm_destroy.argtypes = [H]
    lib.erm_farm_destroy.restype = None
    lib.erm_farm_chains.argtypes = [H]
    lib.erm_farm_chains.restype = C.c_int32
    lib.erm_farm_engine.argtypes = [H, C.c_int32]
    lib.erm_farm_engine.restype = H
    lib.erm_farm_set_data.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.erm_farm_set_state.argtypes = [H, C.c_int32, C.POINTER(erm_state)]
    lib.erm_farm_get_state.argtypes = [H, C.c_int32, C.POINTER(erm_state)]
    lib.erm_farm_run.argtypes = [H, C.c_int64]
    lib.erm_farm_reset_trace.argtypes = [H]
    lib.erm_farm_get_trace.argtypes = [H, C.c_int, C.c_void_p]
    lib.erm_farm_get_mean.argtypes = [H, C.POINTER(erm_state)]
    lib.erm_farm_post_count.argtypes = [H]
    lib.erm_farm_post_count.restype = C.c_int64
    lib.erm_farm_used_rccl.argtypes = [H]
    lib.erm_farm_get_timing.argtypes = [H, C.POINTER(erm_farm_timing), C.c_void_p]
    lib.erm_get_dic.argtypes = [H, C.c_void_p]
    lib.erm_get_convergence.argtypes = [H, C.c_int, C.c_void_p]
    lib.erm_set_seed.argtypes = [H, C.c_uint64]
    lib.erm_farm_get_dic.argtypes = [H, C.c_void_p]
    lib.erm_farm_set_seed.argtypes = [H, C.c_uint64]
    lib.erm_debug_invwishart.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise ErmError(f"libertirt error {rc}: {load().erm_last_error().decode()}")


STATE_FIELDS = ("theta", "a", "b", "zeta", "lambda_", "sig2t", "beta", "sigp", "rho", "nu")


def state_struct(arrays: dict):
    """Build an erm_state from {field: float64 contiguous ndarray or None}; returns (struct, keepalive)."""
    st = erm_state()
    keep = []
    for f in STATE_FIELDS:
        v = arrays.get(f)
        if v is None:
            setattr(st, f, None)
        else:
            assert v.dtype == np.float64 and (v.flags["C_CONTIGUOUS"] or v.flags["F_CONTIGUOUS"])
            keep.append(v)
            setattr(st, f, v.ctypes.data_as(_DP))
    return st, keep


class Engine:
    """Thin RAII wrapper over an erm_handle."""

    def __init__(self, **kw):
        lib = load()
        cfg = erm_config()
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown config field {k}")
            setattr(cfg, k, v)
        self.cfg = cfg
        self._h = C.c_void_p()
        check(lib.erm_create(C.byref(cfg), C.byref(self._h)))
        self._lib = lib

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.erm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_shard(self, rank: int, count: int, n_subj_total: int, row_base: int, exchange):
        """Make this engine one shard of a subject-sharded chain (include/ertirt.h, erm_set_shard).  `exchange(send_ptr, recv_ptr,
        nbytes)` is the all-gather over the shards (see parallel.TorchExchange / parallel.ThreadExchange); an exception it raises
        is kept in `self.exchange_error` and surfaces as an ErmError from the call that triggered the exchange."""
        self.exchange_error = None

        def _cb(user, send, recv, nbytes):
            try:
                exchange(send, recv, int(nbytes))
                return 0
            except BaseException as e:      # never let an exception cross the C frames
                self.exchange_error = e
                return -1

        self._exchange_cb = EXCHANGE_FN(_cb)       # kept alive as long as the engine
        check(self._lib.erm_set_shard(self._h, rank, count, n_subj_total, row_base, self._exchange_cb, None))

    def set_shard_rccl(self, rank: int, count: int, n_subj_total: int, row_base: int, unique_id: bytes):
        """Shard with the library's own in-stream RCCL all-gather (erm_set_shard_rccl); `unique_id`: the 128 bytes of rccl_unique_id()
        made on one rank and handed to all of them."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of rccl_unique_id()")
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(self._lib.erm_set_shard_rccl(self._h, rank, count, n_subj_total, row_base, buf))

    # ---- data / state
    def set_data(self, Y, logT=None, X=None):
        Ya = np.asarray(Y)
        if Ya.dtype != np.bool_ and not np.all((Ya == 0) | (Ya == 1)):
            raise ValueError("Y must contain only 0/1")
        Yf = np.asfortranarray(Ya.astype(np.uint8))
        lt = None if logT is None else np.asfortranarray(logT, dtype=np.float64)
        xx = None if X is None or np.size(X) == 0 else np.asfortranarray(X, dtype=np.float64)
        check(self._lib.erm_set_data(self._h, Yf.ctypes.data, None if lt is None else lt.ctypes.data,
                                     None if xx is None else xx.ctypes.data))

    def simulate_data(self, seed=4321, noise=0, pull_truth=True, **truth):
        """Generate the data set on the device from the true parameters (erm_simulate_data); returns (theta, zeta) of the truth (None with
        pull_truth=False: they stay on the device, erm_get_truth fetches them later if wanted)."""
        arrs = {k: (None if v is None else np.asfortranarray(v, dtype=np.float64)) for k, v in truth.items()}
        st, keep = state_struct(arrs)
        check(self._lib.erm_simulate_data(self._h, C.byref(st), int(seed), int(noise)))
        if not pull_truth:
            return None
        th, ze = np.empty(self.cfg.n_subj), np.empty(self.cfg.n_subj)
        check(self._lib.erm_get_truth(self._h, th.ctypes.data, ze.ctypes.data))
        return th, ze

    def get_data(self):
        """The resident data set as (Y uint8 NxJ, logT float64 NxJ or None, X float64 NxF or None), column-major."""
        c = self.cfg
        N, J = c.n_subj, c.n_item
        F = 0 if c.model in (MODEL_CROSSQR, MODEL_CROSS, MODEL_NULL) else c.n_feat
        Y = np.empty((N, J), dtype=np.uint8, order="F")
        logT = None if c.model == MODEL_MLIRT else np.empty((N, J), dtype=np.float64, order="F")
        X = np.empty((N, F), dtype=np.float64, order="F") if F > 0 else None
        check(self._lib.erm_get_data(self._h, Y.ctypes.data, None if logT is None else logT.ctypes.data, None if X is None else X.ctypes.data))
        return Y, logT, X

    def set_state(self, **arrays):
        arrs = {k: (None if v is None else np.asfortranarray(v, dtype=np.float64)) for k, v in arrays.items()}
        st, keep = state_struct(arrs)
        check(self._lib.erm_set_state(self._h, C.byref(st)))

    def _state_buffers(self, which=None):
        c = self.cfg
        N, J, F = c.n_subj, c.n_item, c.n_feat
        nb = {MODEL_MLIRT: F + 1, MODEL_RTIRT: 2 * (F + 1), MODEL_LATENTQR: F + 2, MODEL_CROSSQR: 0,
              MODEL_NULL: 2 * (F + 1), MODEL_CROSS: 0, MODEL_LATENT: F + 2}[c.model]
        nnu = {MODEL_LATENTQR: N, MODEL_CROSSQR: N * J}.get(c.model, 0)
        sizes = dict(theta=N, a=J, b=J, zeta=N, lambda_=J, sig2t=J, beta=nb, sigp=4, rho=J, nu=nnu)
        return {k: (np.zeros(n, dtype=np.float64) if n and (which is None or k in which) else None) for k, n in sizes.items()}

    def get_state(self, which=None):
        bufs = self._state_buffers(which)
        st, keep = state_struct(bufs)
        check(self._lib.erm_get_state(self._h, C.byref(st)))
        return bufs

    def get_mean(self, which=None):
        bufs = self._state_buffers(which)
        st, keep = state_struct(bufs)
        check(self._lib.erm_get_mean(self._h, C.byref(st)))
        return bufs

    # ---- run / outputs
    def run(self, nsweeps: int):
        check(self._lib.erm_run(self._h, int(nsweeps)))

    def reset_trace(self):
        check(self._lib.erm_reset_trace(self._h))

    def set_seed(self, seed: int):
        """A new seed for the chain's random streams (erm_set_seed): one engine serves every replication of a simulation condition."""
        check(self._lib.erm_set_seed(self._h, int(seed)))
        self.cfg.seed = int(seed)

    def dic(self):
        """erm_get_dic: {Dbar, Dhat, pD, DIC} from device-resident state (the log-likelihood at Post.mean is one evaluation pass on the device)."""
        out = np.empty(4, dtype=np.float64)
        check(self._lib.erm_get_dic(self._h, out.ctypes.data))
        return dict(Dbar=float(out[0]), Dhat=float(out[1]), pD=float(out[2]), DIC=float(out[3]))

    @property
    def rows_done(self):
        return int(self._lib.erm_rows_done(self._h))

    @property
    def post_count(self):
        return int(self._lib.erm_post_count(self._h))

    def trace(self, which: int):
        """Post.ra / rt / qr / logLike as an (nIter, width, nChain) Fortran-ordered array (Julia layout)."""
        w = int(self._lib.erm_trace_width(self._h, which))
        if w <= 0:
            return np.zeros((0,), dtype=np.float64)
        out = np.empty((self.cfg.n_iter, w, self.cfg.n_chain), dtype=np.float64, order="F")
        check(self._lib.erm_get_trace(self._h, which, out.ctypes.data))
        return out

    def diagnostics(self, which: int):
        """(ess, rhat) of every column of Post.ra / rt / qr, computed on the device from the resident traces."""
        w = int(self._lib.erm_trace_width(self._h, which))
        ess, rhat = np.empty(w), np.empty(w)
        check(self._lib.erm_get_diagnostics(self._h, which, ess.ctypes.data, rhat.ctypes.data))
        return ess, rhat

    def convergence(self, which: int):
        """erm_get_convergence: (columns with a defined ESS, of those ESS > 400, columns with a defined R-hat, of those R-hat < 1.1), counted on the device."""
        c = np.zeros(4, dtype=np.int64)
        check(self._lib.erm_get_convergence(self._h, which, c.ctypes.data))
        return tuple(int(v) for v in c)

    def item_trace(self):
        w = int(self._lib.erm_item_trace_width(self._h))
        out = np.empty((self.rows_done, w), dtype=np.float64)
        if out.size:
            check(self._lib.erm_get_item_trace(self._h, out.ctypes.data))
        return out

    def timing(self):
        t = erm_timing()
        check(self._lib.erm_get_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in erm_timing._fields_}


class _Borrowed(Engine):
    """A farm's engine seen through the Engine wrapper (owned by the farm: never destroyed from here)."""

    def __init__(self, lib, handle, cfg):
        self._lib, self._h, self.cfg = lib, handle, cfg

    def close(self):
        self._h = None


class Farm:
    """erm_farm_*: nChain independent chains, chain l on device devices[l] (include/ertirt.h).  The chains sample concurrently on host
    threads of the library; get_mean() is the one collective (RCCL all-reduce over the devices)."""

    def __init__(self, devices, **kw):
        lib = load()
        cfg = erm_config()
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown config field {k}")
            setattr(cfg, k, v)
        self.cfg, self._lib = cfg, lib
        self.devices = [int(d) for d in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        check(lib.erm_farm_create(C.byref(cfg), arr, len(self.devices), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.erm_farm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_chains(self):
        return int(self._lib.erm_farm_chains(self._h))

    def engine(self, chain: int):
        h = self._lib.erm_farm_engine(self._h, int(chain))
        if not h:
            raise ErmError("no such chain")
        c = erm_config.from_buffer_copy(self.cfg)
        c.n_chain, c.chain_id, c.device = 1, int(chain), self.devices[chain]
        return _Borrowed(self._lib, C.c_void_p(h), c)

    def set_data(self, Y, logT=None, X=None):
        Ya = np.asarray(Y)
        if Ya.dtype != np.bool_ and not np.all((Ya == 0) | (Ya == 1)):
            raise ValueError("Y must contain only 0/1")
        Yf = np.asfortranarray(Ya.astype(np.uint8))
        lt = None if logT is None else np.asfortranarray(logT, dtype=np.float64)
        xx = None if X is None or np.size(X) == 0 else np.asfortranarray(X, dtype=np.float64)
        check(self._lib.erm_farm_set_data(self._h, Yf.ctypes.data, None if lt is None else lt.ctypes.data, None if xx is None else xx.ctypes.data))

    def set_state(self, chain: int, **arrays):
        arrs = {k: (None if v is None else np.asfortranarray(v, dtype=np.float64)) for k, v in arrays.items()}
        st, keep = state_struct(arrs)
        check(self._lib.erm_farm_set_state(self._h, int(chain), C.byref(st)))

    def get_state(self, chain: int):
        bufs = self.engine(chain)._state_buffers()
        st, keep = state_struct(bufs)
        check(self._lib.erm_farm_get_state(self._h, int(chain), C.byref(st)))
        return bufs

    def run(self, nsweeps: int):
        check(self._lib.erm_farm_run(self._h, int(nsweeps)))

    def reset_trace(self):
        check(self._lib.erm_farm_reset_trace(self._h))

    @property
    def post_count(self):
        return int(self._lib.erm_farm_post_count(self._h))

    @property
    def used_rccl(self):
        return bool(self._lib.erm_farm_used_rccl(self._h))

    def get_mean(self):
        bufs = self.engine(0)._state_buffers()
        st, keep = state_struct(bufs)
        check(self._lib.erm_farm_get_mean(self._h, C.byref(st)))
        return bufs

    def set_seed(self, seed: int):
        check(self._lib.erm_farm_set_seed(self._h, int(seed)))
        self.cfg.seed = int(seed)

    def dic(self):
        """erm_farm_get_dic: Dbar over the rows of all chains, Dhat at the joint Post.mean (reduced over the devices like get_mean)."""
        out = np.empty(4, dtype=np.float64)
        check(self._lib.erm_farm_get_dic(self._h, out.ctypes.data))
        return dict(Dbar=float(out[0]), Dhat=float(out[1]), pD=float(out[2]), DIC=float(out[3]))

    def timing(self):
        """erm_farm_get_timing: wall-clock of the last run / gather, per-chain device time, ranks of the library's RCCL communicator."""
        t = erm_farm_timing()
        run_ms = np.zeros(self.n_chains, dtype=np.float64)
        check(self._lib.erm_farm_get_timing(self._h, C.byref(t), run_ms.ctypes.data))
        out = {f: getattr(t, f) for f, _ in erm_farm_timing._fields_}
        out["run_ms"] = run_ms
        return out

    def item_trace(self):
        """Item-level trace rows in the single-engine order (row m * nChain + l = iteration m of chain l)."""
        per = [self.engine(l).item_trace() for l in range(self.n_chains)]
        out = np.empty((per[0].shape[0] * len(per), per[0].shape[1]), dtype=np.float64)
        for l, t in enumerate(per):
            out[l::len(per)] = t
        return out

    def trace(self, which: int):
        """Post.ra / rt / qr / logLike as (nIter, width, nChain), chain l in slab l."""
        w = int(self._lib.erm_trace_width(self._lib.erm_farm_engine(self._h, 0), which))
        if w <= 0:
            return np.zeros((0,), dtype=np.float64)
        out = np.empty((self.cfg.n_iter, w, self.n_chains), dtype=np.float64, order="F")
        check(self._lib.erm_farm_get_trace(self._h, which, out.ctypes.data))
        return out


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    check(load().erm_rccl_unique_id(buf))
    return buf.raw


def debug_sample(which, n, par0=None, par1=None, *, seed=1234, site=15, sweep=1, precision=PREC_F64, device=0):
    lib = load()
    out = np.empty(n, dtype=np.float64)
    p0 = None if par0 is None else np.ascontiguousarray(par0, dtype=np.float64)
    p1 = None if par1 is None else np.ascontiguousarray(par1, dtype=np.float64)
    check(lib.erm_debug_sample(device, precision, which, seed, site, sweep, n,
                               None if p0 is None else p0.ctypes.data, None if p1 is None else p1.ctypes.data, out.ctypes.data))
    return out


def sample_gig(p, a, b, n, *, seed=1234, site=15, sweep=1, device=0):
    """n draws of GIG(p, a, b) on the device (erm_sample_gig; rand(GeneralizedInverseGaussian(p, a, b)) of src/GenInvGaussian.jl)."""
    out = np.empty(int(n), dtype=np.float64)
    check(load().erm_sample_gig(device, seed, site, sweep, int(n), float(p), float(a), float(b), out.ctypes.data))
    return out


def debug_invwishart(nu, psi, n, *, seed=1234, sweep=1, device=0):
    """n draws of the structural step's 2 x 2 InverseWishart(nu, Psi) (erm_debug_invwishart): an (n, 2, 2) array."""
    psi = np.ascontiguousarray(np.asarray(psi, dtype=np.float64).reshape(2, 2).T).reshape(4)      # vec(Psi), column-major
    out = np.empty((int(n), 4), dtype=np.float64)
    check(load().erm_debug_invwishart(device, int(seed), int(sweep), int(n), float(nu), psi.ctypes.data, out.ctypes.data))
    return out.reshape(int(n), 2, 2).transpose(0, 2, 1)
