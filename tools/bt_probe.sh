#!/bin/bash
# block sizes tried on small data sets (GPU box): bash tools/bt_probe.sh   (0 = the library's own choice)
for NJ in "1000 15" "1000 50" "10000 50" "30000 50" "100000 15"; do set -- $NJ; for bt in 0 512 1024; do python bench.py --precision ${PREC:-f64} --no-fp32 --cpu-sweeps 0 --nsubj $1 --nitem $2 --steps 1000 --warmup 50 --block-threads $bt 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('N=$1 J=$2 bt=$bt: %.1f us/sweep, %d x %d' % (1e3 * d['ms_per_step'], d['config']['grid_blocks'], d['config']['block_threads']))"; done; done
