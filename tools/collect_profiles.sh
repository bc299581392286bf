#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the default bench command            -> profiles/<tag>_kernel_stats.csv
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots)  -> profiles/<tag>_fetch.csv, _write.csv
#   3. a calibration pass of FETCH_SIZE on a known byte count (row-sum phase only: exactly omega+Y+logT of the workload; needs the
#      -DERM_DIAG_BUILD library variant, built here under gpurun_out/)
#   4. SQ instruction / wait counters                                   -> profiles/<tag>_sq.csv
# then tools/summarize_profiles.py turns them into profiles/traffic.json + profiles/<tag>_summary.md
# usage: bash tools/collect_profiles.sh <tag> [f64|f32]
set -e
TAG=${1:-round2}; PREC=${2:-f64}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}_$PREC; mkdir -p $OUT
DIAG=$PWD/gpurun_out/libertirt_diag.so
if [ ! -f $DIAG ] || [ extendedrtirtmodeling.jl_amd/csrc/erm_kernels.hpp -nt $DIAG ] || [ extendedrtirtmodeling.jl_amd/csrc/ertirt.hip -nt $DIAG ] || [ extendedrtirtmodeling.jl_amd/csrc/erm_rng.hpp -nt $DIAG ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include extendedrtirtmodeling.jl_amd/csrc/ertirt.hip -DERM_DIAG_BUILD -o $DIAG
fi
# 1. is the DEFAULT bench command for that precision (minus the CPU baselines and the other precision's leg); the counter passes
#    serialise kernels, so they use a shorter run
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --precision $PREC --no-fp32 --cpu-sweeps 0 > $OUT/stats.log 2>&1
BENCH="python3 bench.py --precision $PREC --no-fp32 --steps 200 --warmup 20 --cpu-sweeps 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH --no-profile > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH --no-profile > $OUT/write.log 2>&1
ERM_LIB_PATH=$DIAG ERM_TINY_STOP=1 ERM_PASS_STOP=5 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_cal -- $BENCH --no-profile > $OUT/fetch_cal.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq -- $BENCH --no-profile > $OUT/sq.log 2>&1
echo collected
