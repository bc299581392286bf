#!/bin/bash
# SQ instruction counters of the fused sweep kernel for several library builds on one box (diagnostics).
# usage (GPU box, repo root): bash tools/ab_counters.sh libA.so libB.so ...   ("-" = the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  if [ "$L" = "-" ]; then unset ERM_LIB_PATH; else export ERM_LIB_PATH=$PWD/$L; fi
  OUT=gpurun_out/abc/$(basename $L .so); rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -- python3 bench.py --precision f64 --no-fp32 --steps 100 --warmup 10 --cpu-sweeps 0 --no-profile > $OUT/log.txt 2>&1
  python3 - "$OUT" "$L" <<PY
import csv,glob,collections,sys
f=sorted(glob.glob(sys.argv[1]+"/*/*counter_collection.csv"))
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[-1])):
    if "pass_kernel" in r["Kernel_Name"] and ", 0, true" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v[2:])/max(1,len(v[2:]))) for k,v in agg.items()})
PY
done
