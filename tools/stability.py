"""Long-run stability of the fp32 engine: every model for many sweeps on setData*-style data; reports the spread of the traced item
parameters over the second half and fails on any non-finite value.  usage: python tools/stability.py [nsweeps] [nsubj] [nitem]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
J = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for model in ("mlirt", "rtirt", "null", "latentqr", "latent", "crossqr", "cross"):
    Y, logT, X, init, tp = pu.make_problem(model, N, J, 3, seed=11, qRt=0.85)
    dev = pu.run_device(model, Y, logT, X, init, T, precision="f32", qRt=0.85, trace_full=False)
    it = dev["item"][T // 2:]
    ok = np.all(np.isfinite(dev["item"])) and np.all(np.isfinite(dev["ll"]))
    a = it[:, :J].mean(0)
    print(f"{model:9s} finite={ok} rmse(a)={np.sqrt(np.mean((a - tp.a) ** 2)):.3f} sd(a) in [{it[:, :J].std(0).min():.3f}, {it[:, :J].std(0).max():.3f}] "
          f"ll {dev['ll'][0, 0, 0]:.1f} -> {dev['ll'][-1, 0, 0]:.1f}", flush=True)
    assert ok, model
print("stable")
