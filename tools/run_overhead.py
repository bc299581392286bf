"""Wall clock, device time and fixed host overhead of erm_run calls of a few lengths (GPU box, repo root): python tools/run_overhead.py [K ...]
(device us/sweep = the call's event-bracketed device time / K: what the call's bookkeeping launches and the gaps between its graphs add shows as the difference between lengths)"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import bench, __graft_entry__ as ge
import numpy as np
pkg = ge.load_package(); L = pkg._lib
Ks = [int(a) for a in sys.argv[1:]] or [20, 20, 20, 100, 100]
Y, logT, X = bench.make_data(pkg, "rtirt", 100000, 50, 3, seed=1234)
st = bench.init_state("rtirt", 100000, 50, 3, 0)
for prof in (0, 1):
    eng = L.Engine(model=L.MODEL_RTIRT, n_item=50, n_subj=100000, n_feat=3, n_iter=8000, n_chain=1, n_burnin=0, cov2one=1, q_rt=0.85, seed=1234, precision=L.PREC_F64, trace_mode=L.TRACE_FULL, profile=prof)
    eng.set_data(Y, logT, X); eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in st.items()})
    eng.run(200)
    for K in Ks:
        t0 = time.perf_counter(); eng.run(K); t1 = time.perf_counter()
        tm = eng.timing()
        print("profile", prof, "K", K, "wall us/sweep %.2f" % ((t1 - t0) * 1e6 / K), "device us/sweep %.2f" % (tm["run_ms"] * 1e3 / K), "fixed overhead us %.1f" % ((t1 - t0) * 1e6 - tm["run_ms"] * 1e3), flush=True)
