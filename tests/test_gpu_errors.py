"""Error behaviour of the C-ABI boundary on a live GPU: every misuse returns a negative code + message (never a crash, a hang or a
silent fallback), mirroring the reference's own checks where it has them (@assert 0 < qRt < 1 and nu > 0, src/Draw.pl.jl:476-477;
itemtype validation is host-side, tests/test_host_api.py)."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu
L = pu.ge.load_package()._lib


def _engine(model="rtirt", N=200, J=6, F=2, **kw):
    args = dict(model=pu.MODELS.get(model, model), n_item=J, n_subj=N, n_feat=F, n_iter=4, n_chain=1, n_burnin=2, cov2one=1, q_rt=0.85, seed=1,
                precision=1, trace_mode=1)
    args.update(kw)
    return L.Engine(**args)


def _data(N=200, J=6, F=2, seed=0):
    g = np.random.default_rng(seed)
    return (g.random((N, J)) < 0.5).astype(np.uint8), g.normal(3, 0.5, (N, J)), g.standard_normal((N, F))


@pytest.mark.parametrize("kw,msg", [
    (dict(n_item=0), "positive"), (dict(n_subj=0), "positive"), (dict(n_item=897), "n_item too large"), (dict(n_feat=15), "n_feat too large"),
    (dict(model=9), "unknown model"), (dict(precision=7), "precision"), (dict(block_threads=100), "block_threads"),
    (dict(lanes_per_row=3), "lanes_per_row"), (dict(sigp_mode=2), "sigp_mode"), (dict(n_chain=0), "n_chain")])
def test_create_rejects_bad_configurations(kw, msg):
    with pytest.raises(L.ErmError, match=msg):
        _engine(**kw)


@pytest.mark.parametrize("model", ["crossqr", "latentqr"])
@pytest.mark.parametrize("q", [0.0, 1.0, -0.3])
def test_quantile_level_must_be_inside_the_unit_interval(model, q):      # @assert at src/Draw.pl.jl:476
    with pytest.raises(L.ErmError, match="qRt"):
        _engine(model=model, q_rt=q)


def test_set_data_validates_its_inputs():
    Y, logT, X = _data()
    e = _engine()
    with pytest.raises((ValueError, L.ErmError), match="0/1"):      # the ctypes wrapper checks first; the library checks again
        e.set_data(Y * 2, logT, X)
    bad = logT.copy(); bad[3, 2] = np.inf
    with pytest.raises(L.ErmError, match="finite"):
        e.set_data(Y, bad, X)
    Xc = X.copy(); Xc[:, 1] = 2 * Xc[:, 0]
    with pytest.raises(L.ErmError, match="singular"):
        e.set_data(Y, logT, Xc)
    with pytest.raises(L.ErmError, match="logT is required"):
        e.set_data(Y, None, X)
    with pytest.raises(L.ErmError, match="erm_set_data has not been called"):
        e.run(1)
    e.set_data(Y, logT, X)
    e.run(4)


def test_state_and_trace_misuse():
    Y, logT, X = _data()
    e = _engine()
    e.set_data(Y, logT, X)
    with pytest.raises(L.ErmError, match="sig2t must be positive"):
        e.set_state(sig2t=np.zeros(6))
    e.run(2)
    with pytest.raises(L.ErmError, match="trace incomplete"):
        e.trace(L.TRACE_RA)
    with pytest.raises(L.ErmError, match="no post-burn-in"):
        e.get_mean()
    e.run(2)
    assert e.trace(L.TRACE_RA).shape == (4, 200 + 12, 1) and e.post_count == 2
    with pytest.raises(L.ErmError, match="trace capacity exceeded"):
        e.run(1)
    e.reset_trace()
    e.run(4)                                                    # the chain continues after a reset
    s = _engine(trace_mode=0)
    s.set_data(Y, logT, X)
    s.run(4)
    with pytest.raises(L.ErmError, match="ERM_TRACE_FULL"):
        s.trace(L.TRACE_RA)
    assert s.trace(L.TRACE_LOGLIKE).shape == (4, 1, 1) and s.item_trace().shape[0] == 4
    cq = _engine(model="crossqr", F=0)
    cq.set_data(Y, logT, None)
    with pytest.raises(L.ErmError, match="nu must be positive"):      # @assert at src/Draw.pl.jl:477
        cq.set_state(nu=np.zeros((200, 6)))


def test_a_poisoned_state_is_reported_not_propagated():
    """A non-finite parameter sets the sticky device flag: erm_run names the entry instead of returning NaN traces."""
    Y, logT, X = _data()
    e = _engine()
    e.set_data(Y, logT, X)
    e.set_state(theta=np.full(200, np.nan))
    with pytest.raises(L.ErmError, match="non-finite parameter"):
        e.run(4)
