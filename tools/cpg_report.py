#!/usr/bin/env python3
"""Table of tools/cpg_sweep.sh's bench lines (gpurun_out/r4/cpg_*.json): microseconds per chain-sweep and cell-updates/s."""
import glob, json, sys
for f in sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r4") + "/cpg_*.json")):
    t = open(f).read().strip()
    try:
        d = json.loads(t.splitlines()[-1])
        print(f"{f.split('/')[-1][4:-5]:28s} chains {d['n_gpus']}  {d['ms_per_step'] * 1e3 / d['n_gpus']:8.1f} us per chain-sweep  {d['value']:.3e} cell-updates/s")
    except Exception:
        print(f"{f.split('/')[-1]:28s} no line")
