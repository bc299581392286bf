#!/bin/bash
# Instruction-cache counters of the fused sweep kernel (is the 63 KB kernel fetch-bound in its once-per-workgroup tiny step?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/icache; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p -- python3 bench.py --steps 100 --warmup 10 --cpu-sweeps 0 --no-profile "$@" > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections, os
f = sorted(glob.glob("gpurun_out/icache/p/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if ", true, false>" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, sum(v[2:]) / max(1, len(v[2:])))
PY
