#!/bin/bash
# geometry probe at the reference's own sizes (GPU box): bash tools/small_geometry.sh [model]
M=${1:-rtirt}
for NJ in "1000 15" "2000 15" "250 15" "1000 50"; do set -- $NJ
  for G in "0 0" "1024 16" "512 32" "256 64" "1024 32" "512 64" "256 128" "1024 8" "512 16"; do set -- $NJ $G
    python bench.py --model $M --no-fp32 --no-cold --cpu-sweeps 0 --nsubj $1 --nitem $2 --steps 1000 --warmup 50 --block-threads $3 --grid-blocks $4 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('N=$1 J=$2 bt=$3 gb=$4: %.2f us/sweep (%d x %d)' % (1e3 * d['ms_per_step'], d['config']['grid_blocks'], d['config']['block_threads']))
except Exception as e: print('N=$1 J=$2 bt=$3 gb=$4: failed')"
  done
done
