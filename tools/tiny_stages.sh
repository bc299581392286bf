#!/bin/bash
# Stage timing of the tiny step / row pass via the early-return knobs (diagnostics; rocprofv3 kernel averages).
# usage (on the GPU box, repo root): bash tools/tiny_stages.sh TINY "0 1 2 3 4"   |   bash tools/tiny_stages.sh PASS "0 1 5 2 3 4"
KIND=${1:-TINY}; STOPS=${2:-"0 1 2 3 4"}; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in $STOPS; do
  export ERM_${KIND}_STOP=$s
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$KIND$s -o t -- python3 $R/bench.py --steps 100 --warmup 10 --no-profile --cpu-sweeps 0 "$@" > $R/gpurun_out/st_$KIND$s.log 2>&1 </dev/null
  f=$(find $R/gpurun_out/st_$KIND$s -name '*kernel_stats.csv' | head -1)
  echo "stop=$s"
  if [ -n "$f" ]; then cut -c1-160 "$f"; else echo "no stats file"; fi
done
