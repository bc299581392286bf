"""Generates tests/golden/*.npz: small seeded inputs and the CPU oracle's traces for them.
The reference (Julia) cannot run here and holds no golden vectors for this path, so these are BUILD-OWNED regression pins:
they freeze the sampling specification (Philox addressing, sampler algorithms, sweep order) so that neither the oracle nor
the HIP path can drift silently.  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import parity_util as pu  # noqa: E402

CASES = {"mlirt": (160, 7, 3), "rtirt": (160, 7, 3), "latentqr": (160, 7, 3), "crossqr": (160, 7, 0),
         "null": (160, 7, 3), "cross": (160, 7, 0), "latent": (160, 7, 3)}
T = 6

if __name__ == "__main__":
    only = sys.argv[1:]
    for model, (N, J, F) in CASES.items():
        if only and model not in only:
            continue
        Y, logT, X, init, _ = pu.make_problem(model, N, J, 3, seed=21)
        op = pu.OracleProblem(model, Y, logT, X, init, qRt=0.85, cov2one=(model not in ("latentqr", "latent")), seed=1234)
        tr = op.run(T, with_nu=(model == "latentqr"))
        out = dict(Y=Y.astype(np.uint8), T=T, qRt=0.85, seed=1234, ra=tr["ra"], rt=tr["rt"], qr=tr["qr"], ll=tr["ll"])
        if logT is not None:
            out["logT"] = logT
        if X is not None:
            out["X"] = X
        for k, v in init.items():
            out["init_" + k] = np.asarray(v)
        np.savez_compressed(os.path.join(HERE, f"{model}.npz"), **out)
        print(model, {k: np.shape(v) for k, v in out.items()})
