"""The N > 1 path on CPU: two ranks (gloo, 127.0.0.1), each holding the posterior summary of its own chain (produced here by
the oracle as the per-rank engine), combined by the same all-reduce the GPU farm uses over RCCL.  The combined mean must equal
mean(..., dims=(1,3)) over the union of both chains' post-burn-in rows (src/GibbsRtIrt.pl.jl:327-343)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

import parity_util as pu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _chain(rank, N, J, T):
    Y, logT, X, init, _ = pu.make_problem("rtirt", N, J, 3, seed=9)
    g = np.random.default_rng(100 + rank)
    init = dict(init, theta=g.standard_normal(N), zeta=g.standard_normal(N))
    op = pu.OracleProblem("rtirt", Y, logT, X, init, qRt=0.5, chain=rank)
    return op.run(T)


def _worker(rank, world, port, N, J, T, out):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = pu.ge.load_package()
    tr = _chain(rank, N, J, T)
    burn = T // 2
    ra, rt, qr = tr["ra"][burn:].mean(0), tr["rt"][burn:].mean(0), tr["qr"][burn:].mean(0)
    P = pkg.InputPara(theta=ra[:N], a=ra[N:N + J], b=ra[N + J:], zeta=rt[:N], lam=rt[N:N + J], sig2t=rt[N + J:], beta=qr[:8], Sigp=qr[8:])
    comb = pkg.parallel.gather_posterior_summaries(P, T - burn, float(tr["ll"][burn:].sum()))
    if rank == 0:
        np.savez(out, **{k: v for k, v in comb.items()})
    dist.destroy_process_group()


def test_two_rank_chain_farm_summary(tmp_path):
    N, J, T = 120, 6, 12
    out = str(tmp_path / "comb.npz")
    mp.spawn(_worker, args=(2, _free_port(), N, J, T, out), nprocs=2, join=True)
    comb = np.load(out)
    trs = [_chain(r, N, J, T) for r in range(2)]
    burn = T // 2
    ra = np.concatenate([t["ra"][burn:] for t in trs]).mean(0)
    rt = np.concatenate([t["rt"][burn:] for t in trs]).mean(0)
    qr = np.concatenate([t["qr"][burn:] for t in trs]).mean(0)
    assert int(comb["count"]) == 2 * (T - burn)
    assert np.allclose(comb["theta"], ra[:N], rtol=1e-12) and np.allclose(comb["b"], ra[N + J:], rtol=1e-12)
    assert np.allclose(comb["zeta"], rt[:N], rtol=1e-12) and np.allclose(comb["sig2t"], rt[N + J:], rtol=1e-12)
    assert np.allclose(comb["beta"], qr[:8], rtol=1e-12, atol=1e-15) and np.allclose(comb["Sigp"], qr[8:], rtol=1e-12)
    assert np.isclose(float(comb["loglike_sum"]), sum(t["ll"][burn:].sum() for t in trs))
    assert not np.allclose(trs[0]["ra"], trs[1]["ra"])          # the two ranks really ran different chains


def test_single_process_gather_is_identity():
    pkg = pu.ge.load_package()
    P = pkg.InputPara(theta=[1.0, 2.0], a=[0.5], b=[0.1], Sigp=np.eye(2))
    c = pkg.parallel.gather_posterior_summaries(P, 7, 3.5)
    assert c["count"] == 7 and np.allclose(c["theta"], [1, 2]) and np.allclose(c["Sigp"], [1, 0, 0, 1]) and c["loglike_sum"] == 3.5


# ------------------------------------------------------------------------------------------- subject-sharded chains (erm_set_shard)
class _HostCopy:
    """Stands in for libertirt's erm_copy on a box without a GPU: the exchange only ever asks for byte copies."""

    @staticmethod
    def erm_copy(dst, src, nbytes):
        import ctypes
        ctypes.memmove(dst, src, nbytes)
        return 0


def _exchange_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = pu.ge.load_package()
    ex = pkg.parallel.TorchExchange(_HostCopy, device=None)
    got = []
    for width in (7, 300, 7):                       # the statistics rows of different passes have different widths; buffers are reused
        send = np.arange(width, dtype=np.float64) + 1000.0 * rank + width
        recv = np.full(width * world, -1.0)
        ex(send.ctypes.data, recv.ctypes.data, send.nbytes)
        got.append(recv)
    if rank == 1:
        np.savez(out, *got)
    dist.destroy_process_group()


def test_two_rank_statistics_exchange(tmp_path):
    out = str(tmp_path / "ex.npz")
    mp.spawn(_exchange_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    for k, width in enumerate((7, 300, 7)):
        want = np.concatenate([np.arange(width) + 1000.0 * r + width for r in range(2)])
        np.testing.assert_array_equal(got[f"arr_{k}"], want)


def test_shard_rows_cover_the_subjects():
    pkg = pu.ge.load_package()
    for n, c in ((10, 3), (7, 7), (100000, 8), (5, 1)):
        rows = pkg.parallel.shard_rows(n, c)
        assert len(rows) == c and rows[0][0] == 0 and sum(k for _, k in rows) == n
        assert all(rows[r][0] + rows[r][1] == rows[r + 1][0] for r in range(c - 1))
        assert max(k for _, k in rows) - min(k for _, k in rows) <= 1
