#!/bin/bash
# A/B of library builds on ONE box (box-to-box spread on the pool is +-1.5 %): alternating runs of the default bench.
# usage (on the GPU box, from the repo root): bash tools/ab_bench.sh [bench args ...] -- libA.so libB.so ...   ("-" = the in-tree library)
# NB "-" is the working tree AS SNAPSHOT: bench.py rebuilds the in-tree library when a source is newer than it, so an edited tree is measured as edited --
# build the edit as a variant (tools/build_variant.sh), put the tree back, then compare "-" with the variant.
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
