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
