"""Chain farm across the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on
ROCm, "gloo" on CPU for tests), one independent chain per rank, no communication while sampling, and ONE reduction of the
posterior summaries at the end.

The reference's nChain > 1 is a single interleaved Markov chain (`for m in 1:nIter, l in 1:nChain` over one shared Para,
/root/reference/src/GibbsRtIrt.pl.jl:284-289) whose Post.mean averages over iterations and chains jointly (:327-343).
Farming independent chains (rank r uses random stream chain_id = r) leaves that average unchanged in expectation; the
combined mean below is the sample-count weighted average of the per-rank Post.mean vectors, which is exactly
`mean(..., dims=(1,3))` over the union of the ranks' post-burn-in rows.
"""
from __future__ import annotations

import numpy as np

SUMMARY_FIELDS = ("theta", "a", "b", "zeta", "lam", "sig2t", "beta", "Sigp", "rho", "nu")
MAX_CHAINS = 256       # the random streams carry eight bits of the chain id (include/ertirt.h, erm_config.chain_id): chain 256 would replay chain 0


def rank_chain_id(rank: int) -> int:
    """The random stream of the chain a rank runs: its rank, which must fit the eight bits the streams carry (erm_create refuses anything else)."""
    rank = int(rank)
    if not 0 <= rank < MAX_CHAINS:
        raise ValueError(f"rank {rank} cannot be a chain id: independent chains are numbered 0 .. {MAX_CHAINS - 1}")
    return rank


def pack_summary(mean, count: int, loglike_sum: float = 0.0, fields=SUMMARY_FIELDS):
    """Flatten a Post.mean (InputPara) into [count, loglike_sum, count*field0..., count*field1...] (float64)."""
    parts = [np.array([float(count), float(loglike_sum)])]
    layout = []
    for f in fields:
        v = np.asarray(getattr(mean, f), dtype=np.float64).reshape(-1, order="F")
        layout.append((f, v.size))
        parts.append(v * float(count))
    return np.concatenate(parts), layout


def unpack_summary(buf: np.ndarray, layout):
    count = buf[0]
    out = {"count": int(round(count)), "loglike_sum": float(buf[1])}
    o = 2
    for f, n in layout:
        out[f] = buf[o:o + n] / count
        o += n
    return out


def gather_posterior_summaries(mean, count: int, loglike_sum: float = 0.0, *, group=None, device=None):
    """All-reduce (sum) of the count-weighted summaries over the process group; every rank returns the combined means.
    With the nccl backend the buffer must live on the rank's GPU (`device`); with gloo it stays on the CPU."""
    import torch
    import torch.distributed as dist

    buf, layout = pack_summary(mean, count, loglike_sum)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return unpack_summary(t.cpu().numpy(), layout)


# ---------------------------------------------------------------------------------------------------------------------
# Subject sharding of ONE chain (second bullet; include/ertirt.h erm_set_shard)
# ---------------------------------------------------------------------------------------------------------------------
def shard_rows(n_subj: int, count: int):
    """Contiguous, near-equal row ranges [(row_base, n_local)] of `count` shards."""
    base, rem = divmod(int(n_subj), int(count))
    out, lo = [], 0
    for r in range(count):
        n = base + (1 if r < rem else 0)
        out.append((lo, n))
        lo += n
    return out


class TorchExchange:
    """The all-gather erm_set_shard asks for, over a torch.distributed group: RCCL ("nccl") moves the statistics row GPU to GPU over
    xGMI; with "gloo" (CPU tests, or several ranks sharing one GPU) the row is staged through host memory.  The library's buffers are
    copied into / out of torch-owned tensors with erm_copy, so no raw pointer is ever wrapped as a tensor."""

    def __init__(self, lib, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.lib, self.group, self.dist, self.torch = lib, group, dist, torch
        self.world = dist.get_world_size(group)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self._buf = {}

    def __call__(self, send_ptr, recv_ptr, nbytes):
        torch = self.torch
        n = nbytes // 8
        if n not in self._buf:
            self._buf[n] = (torch.empty(n, dtype=torch.float64, device=self.device), torch.empty(n * self.world, dtype=torch.float64, device=self.device))
        ts, tr = self._buf[n]
        if self.lib.erm_copy(ts.data_ptr(), send_ptr, nbytes) != 0:
            raise RuntimeError("erm_copy (send) failed")
        self.dist.all_gather_into_tensor(tr, ts, group=self.group)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if self.lib.erm_copy(recv_ptr, tr.data_ptr(), nbytes * self.world) != 0:
            raise RuntimeError("erm_copy (recv) failed")


class ThreadExchange:
    """In-process all-gather for `count` engines driven by `count` host threads (tests, or one process feeding several GPUs):
    rank r parks its row in a host slot, a barrier, everyone copies the concatenation back."""

    def __init__(self, lib, count: int, timeout: float = 120.0):
        import threading
        self.lib, self.count = lib, count
        self.slots = [None] * count
        self.barrier = threading.Barrier(count, timeout=timeout)

    def for_rank(self, rank: int):
        def exchange(send_ptr, recv_ptr, nbytes):
            host = np.empty(nbytes // 8, dtype=np.float64)
            if self.lib.erm_copy(host.ctypes.data, send_ptr, nbytes) != 0:
                raise RuntimeError("erm_copy (send) failed")
            self.slots[rank] = host
            self.barrier.wait()
            allrows = np.concatenate(self.slots)
            self.barrier.wait()           # nobody overwrites a slot before everyone has read it
            if self.lib.erm_copy(recv_ptr, allrows.ctypes.data, nbytes * self.count) != 0:
                raise RuntimeError("erm_copy (recv) failed")
        return exchange

    def abort(self):
        self.barrier.abort()


def run_sharded_threads(make_engine, count: int, n_subj: int, Y, logT, X, state, nsweeps: int):
    """Drive `count` shards of one chain from `count` host threads of this process.  make_engine(n_local) -> _lib.Engine (not yet
    sharded, no data); Y / logT / X / state['theta'|'zeta'|'nu'] are split by rows.  Returns the engines (data resident, nsweeps run)."""
    import threading
    from . import _lib
    lib = _lib.load()
    ex = ThreadExchange(lib, count)
    rows = shard_rows(n_subj, count)
    engines, errors = [None] * count, [None] * count

    def cut(a, lo, n):
        return None if a is None else np.ascontiguousarray(np.asarray(a)[lo:lo + n])

    def work(r):
        try:
            lo, n = rows[r]
            eng = make_engine(n)
            engines[r] = eng
            eng.set_shard(r, count, n_subj, lo, ex.for_rank(r))
            eng.set_data(cut(Y, lo, n), cut(logT, lo, n), cut(X, lo, n))
            st = dict(state)
            for k in ("theta", "zeta"):
                if st.get(k) is not None:
                    st[k] = cut(st[k], lo, n)
            if st.get("nu") is not None:
                st["nu"] = np.asfortranarray(cut(np.asarray(st["nu"]).reshape(n_subj, -1, order="F"), lo, n)).reshape(-1, order="F")
            eng.set_state(**st)
            eng.run(nsweeps)
        except BaseException as e:
            errors[r] = getattr(engines[r], "exchange_error", None) or e
            ex.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(count)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    return engines
