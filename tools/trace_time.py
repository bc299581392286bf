import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import parity_util as pu
Y, logT, X, init, tp = pu.make_problem("rtirt", 100000, 50, 3, seed=1, qRt=0.5)
for prec in ("f64", "f32"):
    t0 = time.perf_counter()
    r = pu.run_device("rtirt", Y, logT, X, init, 500, precision=prec, qRt=0.5, trace_full=False)
    t1 = time.perf_counter()
    L = pu.ge.load_package()._lib
    eng = L.Engine(model=1, n_item=50, n_subj=100000, n_feat=3, n_iter=500, n_chain=1, n_burnin=250, cov2one=1, q_rt=0.5, seed=1234, precision={"f32": 0, "f64": 1}[prec], trace_mode=1)
    eng.set_data(Y, logT, X); eng.set_state(**init)
    t2 = time.perf_counter(); eng.run(500); t3 = time.perf_counter()
    ra = eng.trace(L.TRACE_RA); t4 = time.perf_counter()
    rt = eng.trace(L.TRACE_RT); t5 = time.perf_counter()
    m = eng.get_mean(); t6 = time.perf_counter()
    print(prec, "run %.3f s, trace RA %.3f s, RT %.3f s, mean %.3f s" % (t3 - t2, t4 - t3, t5 - t4, t6 - t5), ra.shape)
# erm_set_data (InputData -> resident buffers): host-side transposition + constants + upload
for (n, j) in ((100000, 50), (500000, 100)):
    Yb, lTb, Xb, initb, _ = pu.make_problem("rtirt", n, j, 3, seed=2, qRt=0.5)
    for prec in ("f64", "f32"):
        eng = L.Engine(model=1, n_item=j, n_subj=n, n_feat=3, n_iter=4, n_chain=1, n_burnin=2, cov2one=1, q_rt=0.5, seed=1234, precision={"f32": 0, "f64": 1}[prec], trace_mode=0)
        t0 = time.perf_counter(); eng.set_data(Yb, lTb, Xb); t1 = time.perf_counter()
        print("set_data %d x %d %s: %.3f s" % (n, j, prec, t1 - t0))
