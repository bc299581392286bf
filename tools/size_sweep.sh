#!/bin/bash
# Throughput of the GibbsRtIrt engine (default: the fp64 headline engine; PREC=f32 for the fast mode) across data-set sizes (one GPU);
# prints one line per (nSubj, nItem).
PREC=${PREC:-f64}
for N in 1000 10000 100000 500000 2000000; do
  for J in 15 50 100; do
    if [ $((N * J)) -gt 100000000 ]; then continue; fi
    S=1000; [ $N -ge 500000 ] && S=200
    timeout -k 10 200 python bench.py --precision $PREC --no-fp32 --cpu-sweeps 0 --nsubj $N --nitem $J --steps $S --warmup 50 --trace summary > gpurun_out/sz.json 2>/dev/null
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/sz.json").read().strip().splitlines()[-1])
print("N=%8d J=%3d  %9.1f us/sweep  %.3g cell-updates/s  frac %.3f  grid %d x %d" % ($N, $J, d["ms_per_step"]*1e3, d["value"], d["roofline"]["frac"], d["config"]["grid_blocks"], d["config"]["block_threads"]))
PY
  done
done
