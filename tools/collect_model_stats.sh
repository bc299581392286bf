#!/bin/bash
# rocprofv3 kernel statistics of the other samplers' default bench runs (the headline's are collected by collect_profiles.sh)
TAG=${1:-round1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in mlirt latentqr crossqr; do
  OUT=gpurun_out/$TAG/stats_$m; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --cpu-sweeps 0 --model $m > gpurun_out/$TAG/stats_$m.log 2>&1
  f=$(ls -t $OUT/*/*_kernel_stats.csv | head -1); cp $f gpurun_out/$TAG/kernel_stats_$m.csv; head -4 $f | cut -c1-160
done
