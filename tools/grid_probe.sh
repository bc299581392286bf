#!/bin/bash
# workgroup counts tried on one configuration (GPU box): bash tools/grid_probe.sh <nsubj> <nitem> <grid> [<grid> ...]   (0 = the library's own choice)
N=$1; J=$2; shift 2
for g in "$@"; do
  python bench.py --precision ${PREC:-f64} --no-fp32 --cpu-sweeps 0 --nsubj $N --nitem $J --steps 200 --warmup 20 --grid-blocks $g 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); print('grid arg $g: %.1f us/sweep, %d workgroups, %d B LDS' % (1e3 * d['ms_per_step'], d['config']['grid_blocks'], d['config']['lds_bytes']))
except Exception: print('grid arg $g: ERR', l[:200])"
done
