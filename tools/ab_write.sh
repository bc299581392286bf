#!/bin/bash
# WRITE_SIZE / FETCH_SIZE per launch of the fused sweep kernel for library variants (one box): bash tools/ab_write.sh [bench args] -- libA libB ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
for L in "$@"; do
  if [ "$L" = "-" ]; then unset ERM_LIB_PATH; else export ERM_LIB_PATH=$PWD/$L; fi
  for C in WRITE_SIZE FETCH_SIZE; do
    N=$(basename $L .so); [ "$L" = "-" ] && N=intree; D=gpurun_out/abw/${N}_$C; rm -rf $D
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 bench.py --no-fp32 --steps 60 --warmup 10 --cpu-sweeps 0 --no-profile "${ARGS[@]}" > $D.log 2>&1
    python3 - "$D" "$L" "$C" <<PY
import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+"/*/*counter_collection.csv"))
if not f: print("no counter file under", sys.argv[1], "(see", sys.argv[1]+".log)"); sys.exit(0)
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f[-1])) if ", true, false>" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[3]]
print("%-45s %s %.0f KB per launch" % (sys.argv[2], sys.argv[3], sum(v[2:])/max(1,len(v[2:]))))
PY
  done
done
