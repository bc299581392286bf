"""Host-side mirror of the reference's interface (no GPU needed): containers, setCond quirks, sample! argument validation
and error text, the C-ABI library loads and exports every symbol include/ertirt.h declares, struct layouts agree."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import parity_util as pu

pkg = pu.ge.load_package()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(pu.ROOT, "include", "ertirt.h")).read()
    declared = set(re.findall(r"\b(erm_[a-z_]+)\s*\(", hdr))
    lib = pkg._lib.load()
    assert declared == set(pkg._lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.erm_version()


def test_ctypes_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof of every struct of include/ertirt.h as gcc lays them out, against the ctypes mirrors field by field."""
    import subprocess
    L = pkg._lib
    structs = {"erm_config": L.erm_config, "erm_state": L.erm_state, "erm_timing": L.erm_timing, "erm_farm_timing": L.erm_farm_timing}
    alias = {"lambda_": "lambda"}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "ertirt.h"', 'int main(void) {']
    for name, cls in structs.items():
        src.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for f, _ in cls._fields_:
            src.append(f'  printf("{name}.{f} %zu\\n", offsetof({name}, {alias.get(f, f)}));')
    src += ['  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(pu.ROOT, "include"), str(c), "-o", str(exe)], check=True)
    got = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert C.sizeof(cls) == int(got[name]), name
        for f, _ in cls._fields_:
            assert getattr(cls, f).offset == int(got[f"{name}.{f}"]), (name, f)
    assert C.sizeof(L.erm_config) == 112 and L.erm_config.q_rt.offset == 56 and L.erm_config.nu_trace_max_gb.offset == 104


def test_create_without_gpu_reports_an_error_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg._lib.ErmError):
        pkg._lib.Engine(model=1, n_item=5, n_subj=10, n_feat=1, n_iter=2, n_chain=1, n_burnin=1)


def test_setcond_defaults_and_forced_burnin():
    c = pkg.setCond()
    assert (c.nSubj, c.nItem, c.nFeat, c.nIter, c.nChain, c.nThin, c.nRep, c.qRa, c.qRt) == (2000, 15, 3, 5000, 4, 1, 10, 0.5, 0.5)
    assert c.nBurnin == 2500
    assert pkg.setCond(nIter=501, nBurnin=7).nBurnin == 250          # kwarg ignored; round half to even like Julia (250.5 -> 250)
    assert pkg.setCond(nIter=503).nBurnin == 252                      # 251.5 -> 252


def test_input_data_derives_kappa_and_logT():
    Y = np.array([[1, 0], [0, 1], [1, 1]], dtype=bool)
    T = np.exp(np.array([[1.0, 2.0], [3.0, 4.0], [0.5, 0.25]]))
    D = pkg.InputData(Y=Y, T=T, X=np.ones((3, 1)))
    assert np.array_equal(D.κ, Y - 0.5) and np.allclose(D.logT, np.log(T))
    P = pkg.InputPara(θ=[1, 2], Σp=np.eye(2))
    assert np.array_equal(P.theta, [1, 2]) and np.array_equal(getattr(P, "σ²t"), []) and P.Sigp.shape == (2, 2)
    with pytest.raises(AttributeError):
        P.nonsense = 1


def _toy(cls, **kw):
    Cond = pkg.setCond(nSubj=30, nItem=4, nFeat=2, nIter=6, nChain=2, qRt=0.85)
    g = np.random.default_rng(0)
    D = pkg.InputData(Y=g.random((30, 4)) < 0.5, T=np.exp(g.normal(1, 0.3, (30, 4))), X=g.standard_normal((30, 2)))
    return cls(Cond, Data=D, **kw)


def test_constructors_initialise_para_like_the_reference():
    M = _toy(pkg.GibbsRtIrt)
    assert M.Para.theta.shape == (30,) and M.Para.zeta.shape == (30,) and M.Para.beta.shape == (3, 2)
    assert np.all(M.Para.a == 1) and np.all(M.Para.b == 0) and np.all(M.Para.lam == 0) and np.all(M.Para.sig2t == 1)
    assert np.array_equal(M.Para.Sigp, np.eye(2))
    assert _toy(pkg.GibbsMlIrt).Para.beta.shape == (3,) and _toy(pkg.GibbsMlIrt).Para.Sigp.size == 0
    assert _toy(pkg.GibbsRtIrtLatentQr).Para.beta.shape == (4,)
    assert _toy(pkg.GibbsRtIrtCrossQr).Para.rho.shape == (4,)
    assert pkg.GibbsRtIrtQuantile is pkg.GibbsRtIrtLatentQr
    # same seed -> same initial values; different chain_id -> different
    assert np.array_equal(_toy(pkg.GibbsRtIrt, seed=5).Para.theta, _toy(pkg.GibbsRtIrt, seed=5).Para.theta)
    assert not np.array_equal(_toy(pkg.GibbsRtIrt, seed=5).Para.theta, _toy(pkg.GibbsRtIrt, seed=5, chain_id=1).Para.theta)


def test_sample_validates_itemtype_with_the_reference_error_text():
    M = _toy(pkg.GibbsRtIrt)
    with pytest.raises(ValueError, match="Invalid input: the item type must be '1pl' or '2pl'."):
        pkg.sample_b(M, itemtype="3pl")
    with pytest.raises(TypeError):
        pkg.sample_b(_toy(pkg.GibbsRtIrtCrossQr), intercept=True)     # sample!(::GibbsRtIrtCrossQr) has no intercept kwarg
    assert pkg.sample is pkg.sample_b


def test_simtools_generators_shapes_and_ranges():
    Cond = pkg.setCond(nSubj=500, nItem=6, nFeat=3, nIter=4, nChain=1)
    for tp_fn, d_fn in ((pkg.setTrueParaRtIrt, pkg.setDataRtIrt), (pkg.setTrueParaRtIrtCross, pkg.setDataRtIrtCross),
                        (pkg.setTrueParaRtIrtLatent, pkg.setDataRtIrtLatent)):
        tp = tp_fn(Cond, seed=1)
        D = d_fn(Cond, tp, seed=2)
        assert D.Y.shape == (500, 6) and set(np.unique(D.Y)) <= {0, 1} and D.logT.shape == (500, 6) and np.all(np.isfinite(D.logT))
        assert np.all(tp.a > 0) and tp.theta.shape == (500,)
    D = pkg.setDataRtIrt(Cond, pkg.setTrueParaRtIrt(Cond, seed=1), seed=2)
    assert D.logT.min() >= 0                                            # truncated at 0, src/SimTools.jl:169
    tp = pkg.setTrueParaMlIrt(Cond, seed=1)
    D = pkg.setDataMlIrt(Cond, tp, seed=2)
    assert set(np.unique(D.X[:, 0])) <= {0.0, 1.0}
    assert pkg.getRmse([1, 2], [1, 4]) == pytest.approx(np.sqrt(2)) and pkg.getBias([1, 2], [1, 4]) == -1


def test_ess_rhat_estimator_on_known_processes():
    """Host twin of the device diagnostics kernel: iid draws have ESS ~ number of draws and R-hat ~ 1; an AR(1) process with
    coefficient phi has ESS ~ n (1 - phi) / (1 + phi); chains with different means have R-hat >> 1; a constant column is NaN."""
    g = np.random.default_rng(0)
    T, C = 4000, 2
    ess, rhat = pkg.ess_rhat(g.standard_normal((T, C)))
    assert 0.8 * T * C < ess < 1.25 * T * C and abs(rhat - 1) < 0.01
    phi = 0.8
    x = np.zeros((T, C))
    e = g.standard_normal((T, C))
    for t in range(1, T):
        x[t] = phi * x[t - 1] + e[t]
    ess, rhat = pkg.ess_rhat(x)
    want = T * C * (1 - phi) / (1 + phi)
    assert 0.7 * want < ess < 1.4 * want and rhat < 1.02
    ess, rhat = pkg.ess_rhat(g.standard_normal((T, C)) + np.array([0.0, 3.0]))
    assert rhat > 1.5 and ess < 20
    assert all(np.isnan(v) for v in pkg.ess_rhat(np.ones((T, C))))


def test_metrics_over_replications():
    """getMetrics / getMetrics2 (src/SimTools.jl:500-551) on a hand-made replication dictionary."""
    true = np.array([1.0, 2.0, 4.0])
    run = {"True": {"a": true}, 1: {"a": true + 0.1}, 2: {"a": true - 0.1}, 3: {"a": true * 1.0}}
    m = pkg.getMetrics(run, par="a")
    assert abs(m["Bias"]) < 1e-12 and abs(m["Rmse"] - np.sqrt(2 * 0.01 / 3)) < 1e-12 and abs(m["Corr"] - 1) < 1e-12
    m2 = pkg.getMetrics2(run, par="a")
    assert abs(m2["relativeBias"]) < 1e-12 and abs(m2["normalizedRmse"] - np.sqrt(2 * 0.01 / 3) / (4.1 - 0.9)) < 1e-12


def test_julia_shim_mirrors_the_header_structs():
    """The Julia shim cannot be executed here (no Julia toolchain), so its ccall struct mirrors are checked statically: ErmConfig /
    ErmState list exactly the fields of erm_config / erm_state in include/ertirt.h, in order and with matching widths, and every
    entry point the shim ccalls is declared in the header."""
    hdr = open(os.path.join(pu.ROOT, "include", "ertirt.h")).read()
    jl = open(os.path.join(pu.ROOT, "extendedrtirtmodeling.jl_amd", "julia", "ExtendedRtIrtModelingAMD.jl")).read()

    def c_fields(name):
        body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} " + name + ";", hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, names = decl.split(None, 1)
            for n in names.split(","):
                n = n.strip()
                out.append((n.lstrip("*"), "ptr" if n.startswith("*") or ctype.endswith("*") else ctype))
        return out

    def jl_fields(name):
        body = re.search(r"struct " + name + r"\b(.*?)\nend", jl, re.S).group(1)
        body = re.sub(r"#.*", "", body)
        return [(m.group(1), m.group(2)) for m in re.finditer(r"(\w+)::([\w{}]+)", body)]

    width = {"int32_t": "Int32", "int64_t": "Int64", "double": "Float64", "uint64_t": "UInt64", "ptr": "Ptr{Float64}"}
    for cname, jname in (("erm_config", "ErmConfig"), ("erm_state", "ErmState")):
        cf, jf = c_fields(cname), jl_fields(jname)
        assert [n for n, _ in cf] == [n for n, _ in jf], (cname, cf, jf)
        assert [width[t] for _, t in cf] == [t for _, t in jf], cname
    called = set(re.findall(r"ccall\(\(:(erm_\w+)", jl))
    declared = set(re.findall(r"\b(erm_[a-z_]+)\s*\(", hdr))
    assert called and called <= declared, called - declared


def test_erm_create_refuses_what_it_cannot_honour_before_touching_a_device():
    """chain ids beyond the eight bits the random streams carry (chain 256 would silently replay chain 0), unknown flag bits (a caller built against
    another header), a negative / NaN nu-trace budget: ERM_ERR_ARG with a message, checked before any HIP call (so this runs without a GPU).  The struct
    layout version is exported for binders."""
    import ctypes as C
    L = pkg._lib
    lib = L.load()
    assert lib.erm_abi_version() == L.ABI_VERSION == int(re.search(r"#define ERM_ABI_VERSION (\d+)", open(os.path.join(pu.ROOT, "include", "ertirt.h")).read()).group(1))
    assert b"0.4" in lib.erm_version()

    def create(**kw):
        cfg = L.erm_config(model=1, n_item=5, n_subj=50, n_feat=2, n_iter=4, n_chain=1, n_burnin=2, q_rt=0.5, seed=1, precision=1, trace_mode=1)
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.erm_create(C.byref(cfg), C.byref(h))
        return rc, lib.erm_last_error().decode()

    for kw, what in ((dict(chain_id=256), "chain_id"), (dict(chain_id=-1), "chain_id"), (dict(flags=64), "flags"), (dict(flags=1 << 20), "flags"),
                     (dict(nu_trace_max_gb=-1.0), "nu_trace_max_gb"), (dict(nu_trace_max_gb=float("nan")), "nu_trace_max_gb"), (dict(trace_mode=7), "trace_mode")):
        rc, msg = create(**kw)
        assert rc == -1 and what in msg, (kw, rc, msg)
    with pytest.raises(ValueError, match="chain_id"):
        pkg.GibbsRtIrt(pkg.setCond(nSubj=10, nItem=3, nFeat=1, nIter=4, nChain=1), chain_id=300)
    with pytest.raises(ValueError):
        pkg.parallel.rank_chain_id(256)
    assert pkg.parallel.rank_chain_id(7) == 7
