#!/bin/bash
# PG-phase attempt / lane-efficiency counters of the default workload (GPU box, repo root): -DERM_DIAG_BUILD library with ERM_PASS_STOP=9
# usage: bash tools/pg_counters.sh [bench args]
R=$GRAFT_REPO_ROOT; DIAG=$R/gpurun_out/libertirt_diag.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I $R/include $R/extendedrtirtmodeling.jl_amd/csrc/ertirt.hip -DERM_DIAG_BUILD -DERM_DIAG_COUNTERS -o $DIAG || exit 1
for P in f64 f32; do
  ERM_LIB_PATH=$DIAG ERM_PASS_STOP=9 python3 $R/bench.py --precision $P --no-fp32 --no-cold --cpu-sweeps 0 --steps 200 --warmup 20 --clock-warmup-ms 0 "$@" 2>&1 | grep "erm dbg" | tail -1
done
