#!/bin/bash
# A/B of library builds on ONE box (box-to-box spread on the pool is +-1.5 %): alternating runs of the default bench.
# usage (on the GPU box, from the repo root): bash tools/ab_bench.sh [bench args ...] -- libA.so libB.so ...   ("-" = the in-tree library)
ARGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
for k in 1 2 3; do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset ERM_LIB_PATH; else export ERM_LIB_PATH=$PWD/$L; fi
    python bench.py --cpu-sweeps 0 "${ARGS[@]}" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); f=d.get('fp32') or {}
print('%-40s %s us/sweep %.2f kernel %.2f%s' % ('$L', d['dtype'], 1e3*d['ms_per_step'], d['roofline']['launch_us'], ('   f32 %.2f kernel %.2f' % (1e3*f['ms_per_step'], f['roofline']['launch_us'])) if f else ''))"
  done
done
