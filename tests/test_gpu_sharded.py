"""Subject sharding of one chain (second bullet; include/ertirt.h erm_set_shard): `count` engines on the one GPU of the
test box, driven by host threads, exchange one row of statistics per row pass.  The random streams are addressed by the global subject
index, so the sharded chain must reproduce the oracle's UNSHARDED chain up to the summation order of the statistics."""
import numpy as np
import pytest

import __graft_entry__ as ge
import parity_util as pu
from parity_util import MODELS

pytestmark = pytest.mark.gpu


def _sharded(model, N, J, nsweeps, count, *, F=3, precision="f64", seed=7, qRt=0.85, **opts):
    pkg = ge.load_package()
    L = pkg._lib
    Y, logT, X, init, tp = pu.make_problem(model, N, J, F, seed=seed, qRt=qRt)
    cov2one = model not in ("latentqr", "latent")
    Fx = 0 if X is None else X.shape[1]

    def make_engine(n_local):
        return L.Engine(model=MODELS[model], n_item=J, n_subj=n_local, n_feat=Fx, n_iter=nsweeps, n_chain=1, n_burnin=nsweeps // 2,
                        cov2one=int(cov2one), q_rt=qRt, seed=1234, precision={"f32": 0, "f64": 1}[precision], trace_mode=1, **opts)

    st = {("lambda_" if k == "lam" else k): v for k, v in init.items()}
    engines = pkg.parallel.run_sharded_threads(make_engine, count, N, Y, logT, X, st, nsweeps)
    op = pu.OracleProblem(model, Y, logT, X, init, qRt=qRt, cov2one=cov2one, seed=1234)
    orc = op.run(nsweeps, with_nu=(model in ("latentqr", "crossqr")))
    rows = pkg.parallel.shard_rows(N, count)
    ra = [e.trace(L.TRACE_RA)[:, :, 0] for e in engines]
    out = {"orc": orc, "model": model, "engines": engines}
    for r in range(1, count):            # item blocks are identical on every shard, bit for bit
        np.testing.assert_array_equal(ra[r][:, rows[r][1]:], ra[0][:, rows[0][1]:])
    out["dev_ra"] = np.concatenate([ra[r][:, :rows[r][1]] for r in range(count)] + [ra[0][:, rows[0][1]:]], axis=1)
    if model != "mlirt":
        rt = [e.trace(L.TRACE_RT)[:, :, 0] for e in engines]
        for r in range(1, count):
            np.testing.assert_array_equal(rt[r][:, rows[r][1]:], rt[0][:, rows[0][1]:])
        out["dev_rt"] = np.concatenate([rt[r][:, :rows[r][1]] for r in range(count)] + [rt[0][:, rows[0][1]:]], axis=1)
    qr = [e.trace(L.TRACE_QR)[:, :, 0] for e in engines]
    if model == "latentqr":
        k = Fx + 2 + 4
        out["dev_qr"] = np.concatenate([qr[0][:, :k]] + [qr[r][:, k:] for r in range(count)], axis=1)
    elif model == "crossqr":
        k = J + 4
        nus = [qr[r][:, k:].reshape(nsweeps, rows[r][1], J, order="F") for r in range(count)]       # vec(nu) is column-major [n_local x J]
        out["dev_qr"] = np.concatenate([qr[0][:, :k], np.concatenate(nus, axis=1).reshape(nsweeps, N * J, order="F")], axis=1)
    else:
        out["dev_qr"] = qr[0]
    ll = [e.trace(L.TRACE_LOGLIKE)[:, 0, 0] for e in engines]
    for r in range(1, count):
        np.testing.assert_array_equal(ll[r], ll[0])
    out["dev_ll"] = ll[0]
    return out


@pytest.mark.parametrize("model", list(MODELS))
def test_two_shards_reproduce_the_unsharded_oracle_chain(model):
    res = _sharded(model, N=301, J=9, nsweeps=3 if model == "crossqr" else 6, count=2)
    assert pu.max_rel_err(res) < 1e-8


def test_three_uneven_shards_and_many_workgroups():
    res = _sharded("rtirt", N=2000, J=12, nsweeps=5, count=3, block_threads=128, grid_blocks=40)
    assert pu.max_rel_err(res) < 1e-8


def test_one_shard_is_the_two_kernel_schedule():
    res = _sharded("latentqr", N=257, J=7, nsweeps=5, count=1)
    assert pu.max_rel_err(res) < 1e-8


def test_single_rank_rccl_shard_with_graph_replay():
    """erm_set_shard_rccl with a communicator of ONE rank (all a one-GPU box allows): the in-stream all-gather, the fused
    one-launch sweep and hipGraph replay (40 sweeps > one 32-sweep graph) against the oracle."""
    pkg = ge.load_package()
    L = pkg._lib
    N, J, T = 300, 8, 40
    Y, logT, X, init, _ = pu.make_problem("rtirt", N, J, 3, seed=7)
    eng = L.Engine(model=MODELS["rtirt"], n_item=J, n_subj=N, n_feat=3, n_iter=T, n_chain=1, n_burnin=T // 2, cov2one=1, q_rt=0.85, seed=1234,
                   precision=1, trace_mode=1)
    eng.set_shard_rccl(0, 1, N, 0, L.rccl_unique_id())
    eng.set_data(Y, logT, X)
    eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in init.items()})
    eng.run(T)
    orc = pu.OracleProblem("rtirt", Y, logT, X, init, qRt=0.85, cov2one=True, seed=1234).run(T)
    assert pu.rel_err(eng.trace(L.TRACE_RA)[:, :, 0], orc["ra"]).max() < 1e-7
    assert pu.rel_err(eng.trace(L.TRACE_LOGLIKE)[:, 0, 0], orc["ll"]).max() < 1e-8
    with pytest.raises(L.ErmError):
        eng.set_shard_rccl(0, 1, N, 0, L.rccl_unique_id())          # once only, and never after the data


def test_f32_shards_match_oracle_for_one_sweep():
    """Same criteria as test_gpu_parity.test_f32_one_sweep (absolute tolerances; a rare flipped PG decision moves a few rows)."""
    N = 3000
    res = _sharded("rtirt", N=N, J=20, nsweeps=1, count=2, precision="f32")

    def abs_err(d, o):
        return np.abs(d - o) / np.maximum(np.abs(o), 1.0)

    for k in ("ra", "rt"):
        assert np.mean(abs_err(res["dev_" + k][0, :N], res["orc"][k][0, :N]) > 5e-4) < 2e-3
        assert abs_err(res["dev_" + k][0, N:], res["orc"][k][0, N:]).max() < 2e-3
    assert abs(res["dev_ll"][0] - res["orc"]["ll"][0]) < 1e-4 * abs(res["orc"]["ll"][0])


def test_posterior_means_of_shards_concatenate():
    res = _sharded("rtirt", N=400, J=8, nsweeps=8, count=2)
    th = np.concatenate([e.get_mean()["theta"] for e in res["engines"]])
    np.testing.assert_allclose(th, res["orc"]["ra"][4:, :400].mean(axis=0), rtol=1e-8, atol=1e-10)
    a0, a1 = (e.get_mean()["a"] for e in res["engines"])
    np.testing.assert_array_equal(a0, a1)


def test_shard_misuse_is_refused():
    pkg = ge.load_package()
    L = pkg._lib
    eng = L.Engine(model=MODELS["rtirt"], n_item=5, n_subj=50, n_feat=0, n_iter=4, n_chain=1, n_burnin=2, cov2one=1, q_rt=0.85, seed=1, precision=1, trace_mode=1)
    noop = lambda s, r, n: None
    for args in ((2, 2, 100, 0), (0, 0, 100, 0), (0, 2, 40, 0), (0, 2, 100, 60), (0, 2, 2 ** 32, 0)):
        with pytest.raises(L.ErmError):
            eng.set_shard(*args, noop)

    def boom(s, r, n):
        raise RuntimeError("link down")
    eng.set_shard(0, 2, 100, 0, boom)
    rng = np.random.default_rng(0)
    with pytest.raises(L.ErmError, match="exchange"):
        eng.set_data(rng.integers(0, 2, (50, 5)), rng.normal(3, 0.5, (50, 5)), None)
    assert isinstance(eng.exchange_error, RuntimeError)
    with pytest.raises(L.ErmError):
        eng.simulate_data(a=np.ones(5), b=np.zeros(5), lambda_=np.ones(5) * 3, sig2t=np.ones(5))


# ------------------------------------------------------------------------------------------- two processes, one GPU, gloo
def _proc_worker(rank, world, port, N, J, T, out):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    L = pkg._lib
    Y, logT, X, init, _ = pu.make_problem("rtirt", N, J, 3, seed=7)
    lo, n = pkg.parallel.shard_rows(N, world)[rank]
    eng = L.Engine(model=MODELS["rtirt"], n_item=J, n_subj=n, n_feat=3, n_iter=T, n_chain=1, n_burnin=T // 2, cov2one=1, q_rt=0.85, seed=1234,
                   precision=1, trace_mode=1, device=0)
    eng.set_shard(rank, world, N, lo, pkg.parallel.TorchExchange(L.load(), device=None))       # gloo: the row is staged through the host
    eng.set_data(Y[lo:lo + n], logT[lo:lo + n], X[lo:lo + n])
    st = {("lambda_" if k == "lam" else k): v for k, v in init.items()}
    st["theta"], st["zeta"] = st["theta"][lo:lo + n], st["zeta"][lo:lo + n]
    eng.set_state(**st)
    eng.run(T)
    np.savez(out + f".{rank}.npz", ra=eng.trace(L.TRACE_RA)[:, :, 0], ll=eng.trace(L.TRACE_LOGLIKE)[:, 0, 0])
    dist.destroy_process_group()


def test_two_processes_share_one_chain_over_gloo(tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    N, J, T = 500, 10, 5
    out = str(tmp_path / "shard")
    mp.spawn(_proc_worker, args=(2, port, N, J, T, out), nprocs=2, join=True)
    Y, logT, X, init, _ = pu.make_problem("rtirt", N, J, 3, seed=7)
    orc = pu.OracleProblem("rtirt", Y, logT, X, init, qRt=0.85, cov2one=True, seed=1234).run(T)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    np.testing.assert_array_equal(r0["ll"], r1["ll"])
    ra = np.concatenate([r0["ra"][:, :250], r1["ra"][:, :250], r0["ra"][:, 250:]], axis=1)
    assert pu.rel_err(ra, orc["ra"]).max() < 1e-8
    assert pu.rel_err(r0["ll"], orc["ll"]).max() < 1e-8


def test_sampler_classes_shard_like_the_reference_surface():
    """GibbsRtIrt(Cond; Data, shard=...) + sample!: two shards (host threads, ThreadExchange) fill Post exactly like the unsharded sampler
    with the same seed -- item blocks, qr and logLike to 1e-8 (fp64), subject blocks concatenated."""
    import threading
    pkg = ge.load_package()
    N, J, T = 240, 8, 6
    Y, logT, X, _, _ = pu.make_problem("rtirt", N, J, 3, seed=7)
    Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=3, nIter=T, nChain=1)
    ref = pkg.sample_b(pkg.GibbsRtIrt(Cond, Data=pkg.InputData(Y=Y, T=np.exp(logT), X=X), seed=5, precision="f64"))
    ex = pkg.parallel.ThreadExchange(pkg._lib.load(), 2)
    rows = pkg.parallel.shard_rows(N, 2)
    out, err = [None, None], [None, None]

    def work(r):
        try:
            lo, n = rows[r]
            C = pkg.setCond(nSubj=n, nItem=J, nFeat=3, nIter=T, nChain=1)
            D = pkg.InputData(Y=Y[lo:lo + n], T=np.exp(logT[lo:lo + n]), X=X[lo:lo + n])
            out[r] = pkg.sample_b(pkg.GibbsRtIrt(C, Data=D, seed=5, precision="f64", shard=(r, 2, N, lo, ex.for_rank(r))))
        except BaseException as e:
            err[r] = e
            ex.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert err == [None, None], err
    n0 = rows[0][1]
    ra = np.concatenate([out[0].Post.ra[:, :n0], out[1].Post.ra[:, :N - n0], out[0].Post.ra[:, n0:]], axis=1)
    assert pu.rel_err(ra[:, :, 0], ref.Post.ra[:, :, 0]).max() < 1e-8
    assert pu.rel_err(out[0].Post.qr[:, :, 0], ref.Post.qr[:, :, 0]).max() < 1e-8
    assert pu.rel_err(out[1].Post.logLike[:, 0, 0], ref.Post.logLike[:, 0, 0]).max() < 1e-8
    np.testing.assert_array_equal(out[0].Post.mean.a, out[1].Post.mean.a)
