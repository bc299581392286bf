#!/usr/bin/env python3
"""Run-to-run determinism stress (GPU box): the farm test's comparison (3 chains sampled concurrently by the library's host threads against the same chains on
separate engines, bit for bit), repeated.  usage: python tools/determinism_stress.py [reps] [block_threads ...]   (0 = the library's own choice)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import parity_util as pu

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bts = [int(a) for a in sys.argv[2:]] or [0]
L = pu.ge.load_package()._lib
N, J, T, nch = 600, 8, 12, 3
bad = 0
for bt in bts:
    for force in ("0", "1"):
        for model in ("rtirt", "mlirt", "latentqr", "crossqr"):
            if model == "latentqr" and bt > 768: continue
            Y, logT, X, init, _ = pu.make_problem(model, N, J)
            g = np.random.default_rng(5)
            inits = [dict(init, theta=g.standard_normal(N)) for _ in range(nch)]
            kw = dict(model=pu.MODELS[model], n_item=J, n_subj=N, n_feat=0 if X is None else X.shape[1], n_iter=T, n_chain=1, n_burnin=T // 2,
                      cov2one=int(model not in ("latentqr", "latent")), q_rt=0.85, seed=1234, precision=1, trace_mode=1, block_threads=bt)
            fkw = dict(kw, flags=L.FLAG_FARM_FORCE_RCCL if force == "1" else 0)
            ref = [None] * nch
            nsep = nfarm = 0
            for r in range(reps):
                farm = L.Farm([0] * nch, **fkw); farm.set_data(Y, logT, X)
                for l in range(nch): farm.set_state(l, **{("lambda_" if k == "lam" else k): v for k, v in inits[l].items()})
                farm.run(T)
                ft = farm.trace(L.TRACE_RA)
                for l in range(nch):
                    e = L.Engine(chain_id=l, **kw); e.set_data(Y, logT, X)
                    e.set_state(**{("lambda_" if k == "lam" else k): v for k, v in inits[l].items()}); e.run(T)
                    tr = e.trace(L.TRACE_RA)[:, :, 0].copy()
                    if ref[l] is None: ref[l] = tr
                    elif not np.array_equal(ref[l], tr): nsep += 1
                    if not np.array_equal(ft[:, :, l], ref[l]): nfarm += 1
            print(f"block_threads {bt:5d} rccl {force} {model:9s}: separate-engine mismatches {nsep}/{nch * (reps - 1)}, farm mismatches {nfarm}/{nch * reps}", flush=True)
            bad += nsep + nfarm
sys.exit(1 if bad else 0)
