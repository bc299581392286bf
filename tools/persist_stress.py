#!/usr/bin/env python3
"""Run-to-run stress of the persistent schedule (GPU box): a chain of T sweeps in ONE persistent launch, repeated, against the same chain on per-sweep launches
(ERM_FLAG_NO_PERSIST), bit for bit -- the packet exchange between the workgroups of the persistent launch has no barrier to hide a race behind.
usage: python tools/persist_stress.py [reps] [sweeps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import parity_util as pu

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
L = pu.ge.load_package()._lib
bad = 0
for model, N, J in (("rtirt", 1000, 15), ("mlirt", 1000, 15), ("latentqr", 2000, 15), ("rtirt", 37, 5), ("null", 5000, 20), ("latent", 250, 100), ("rtirt", 3000, 40), ("mlirt", 6000, 20)):
    Y, logT, X, init, _ = pu.make_problem(model, N, J)
    for prec in ("f64", "f32"):
        geom = {}
        per = pu.run_device(model, Y, logT, X, init, T, precision=prec, trace_full=False)
        assert per["engine"].timing()["persistent"] == 1, (model, N, J)
        tm = per["engine"].timing()
        ref = pu.run_device(model, Y, logT, X, init, T, precision=prec, trace_full=False, flags=L.FLAG_NO_PERSIST, block_threads=tm["block_threads"], grid_blocks=tm["grid_blocks"])
        n = 0
        for r in range(reps):
            got = pu.run_device(model, Y, logT, X, init, T, precision=prec, trace_full=False)
            if not (np.array_equal(got["item"], ref["item"]) and np.array_equal(got["ll"], ref["ll"])): n += 1
        print(f"{model:9s} {N:5d} x {J:3d} {prec}: {tm['grid_blocks']} x {tm['block_threads']}, {reps} persistent runs of {T} sweeps, mismatches against the per-sweep chain {n}", flush=True)
        bad += n
sys.exit(1 if bad else 0)
