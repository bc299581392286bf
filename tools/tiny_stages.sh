#!/bin/bash
# Stage timing of the tiny step / row pass via the early-return knobs (rocprofv3 kernel averages).  The knobs exist only in a library
# built with -DERM_DIAG_BUILD (they leave GARBAGE results): this script builds that variant under gpurun_out/ and points ERM_LIB_PATH at it.
# Note: an early return leaves a garbage chain, so stages AFTER the stop are not comparable across stops when their cost depends on the
# state (the fp64 Polya-Gamma phase takes its slow path for |eta| > 20); use the -DERM_TIMELINE_BUILD timeline for those.
# usage (on the GPU box, repo root): bash tools/tiny_stages.sh TINY "0 1 2 3 4"   |   bash tools/tiny_stages.sh PASS "0 1 5 2 3 4" [bench.py flags]
KIND=${1:-TINY}; STOPS=${2:-"0 1 2 3 4"}; shift; shift
R=$GRAFT_REPO_ROOT
DIAG=$R/gpurun_out/libertirt_diag.so
if [ ! -f $DIAG ] || [ $R/extendedrtirtmodeling.jl_amd/csrc/erm_kernels.hpp -nt $DIAG ] || [ $R/extendedrtirtmodeling.jl_amd/csrc/ertirt.hip -nt $DIAG ] || [ $R/extendedrtirtmodeling.jl_amd/csrc/erm_rng.hpp -nt $DIAG ]; then
  mkdir -p $R/gpurun_out
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I $R/include $R/extendedrtirtmodeling.jl_amd/csrc/ertirt.hip -DERM_DIAG_BUILD -o $DIAG || exit 1
fi
export ERM_LIB_PATH=$DIAG
cd /tmp && export TMPDIR=/tmp
for s in $STOPS; do
  export ERM_${KIND}_STOP=$s
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$KIND$s -o t -- python3 $R/bench.py --steps 100 --warmup 10 --no-profile --no-fp32 --cpu-sweeps 0 "$@" > $R/gpurun_out/st_$KIND$s.log 2>&1 </dev/null
  f=$(find $R/gpurun_out/st_$KIND$s -name '*kernel_stats.csv' | head -1)
  echo "stop=$s"
  if [ -n "$f" ]; then cut -c1-160 "$f"; else echo "no stats file"; fi
done
