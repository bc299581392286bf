#!/bin/bash
# Dynamic VALU instruction mix of the fused sweep kernel by class (GPU box, repo root): SQ_INSTS_VALU_* in separate rocprofv3 --pmc passes of the bench command,
# plus the measured issue cost of each class (tools/ubench/valu_rate.hip).  tools/valu_weighted.py turns both into the class-weighted VALU-busy fraction.
# usage: bash tools/valu_classes.sh <tag> [f64|f32]   ->  gpurun_out/<tag>_valu_<prec>/{classes.json, valu_rate.txt}
TAG=${1:-round4}; PREC=${2:-f64}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}_valu_$PREC; mkdir -p $OUT
BENCH="python3 bench.py --precision $PREC --no-fp32 --steps 100 --warmup 10 --cpu-sweeps 0 --no-profile --no-cold --clock-warmup-ms 0"
k=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  k=$((k+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$k -- $BENCH > $OUT/p$k.log 2>&1
done
python3 - "$OUT" <<PY
import csv, glob, collections, json, sys
out = sys.argv[1]
agg = {}
for f in sorted(glob.glob(out + "/p*/*/*counter_collection.csv")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pass_kernel" in r["Kernel_Name"] and ", 0, true, false>" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        agg[k] = sum(v[2:]) / max(1, len(v[2:]))
json.dump(agg, open(out + "/classes.json", "w"), indent=1)
print(agg)
PY
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o $OUT/valu_rate && $OUT/valu_rate > $OUT/valu_rate.txt && cat $OUT/valu_rate.txt
