#!/usr/bin/env python3
"""One chain on bench.py's synthetic workload, every sweep launched singly (ERM_FLAG_NO_GRAPH): the program tools/stage_budget.py profiles.
usage: python3 tools/one_chain.py [--model rtirt] [--precision f64] [--nsubj 100000] [--nitem 50] [--sweeps 12]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="rtirt"); ap.add_argument("--precision", default="f64")
ap.add_argument("--nsubj", type=int, default=100000); ap.add_argument("--nitem", type=int, default=50); ap.add_argument("--sweeps", type=int, default=12)
a = ap.parse_args()
pkg = ge.load_package()
L = pkg._lib
Y, logT, X = bench.make_data(pkg, a.model, a.nsubj, a.nitem, 3, seed=1234)
st = bench.init_state(a.model, a.nsubj, a.nitem, 3, 0)
eng = L.Engine(model=getattr(L, "MODEL_" + a.model.upper()), n_item=a.nitem, n_subj=a.nsubj, n_feat=0 if X is None else 3, n_iter=a.sweeps, n_chain=1, n_burnin=0,
               cov2one=int(a.model not in ("latentqr", "latent")), q_rt=0.85, seed=1234, precision=L.PREC_F32 if a.precision == "f32" else L.PREC_F64,
               trace_mode=L.TRACE_FULL, flags=L.FLAG_NO_GRAPH)
eng.set_data(Y, logT, X)
eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in st.items()})
eng.run(a.sweeps)
print("done", eng.timing()["run_ms"])
