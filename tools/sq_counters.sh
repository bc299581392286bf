cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sqx; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 bench.py --precision f64 --no-fp32 --steps 100 --warmup 10 --cpu-sweeps 0 --no-profile > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 bench.py --precision f64 --no-fp32 --steps 100 --warmup 10 --cpu-sweeps 0 --no-profile > $OUT/sq2.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("sq","sq2"):
    f=sorted(glob.glob("gpurun_out/sqx/%s/*/*counter_collection.csv"%d))
    if not f: print("no file",d); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[-1])):
        if ", true, false>" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, sum(v[2:])/max(1,len(v[2:])))
PY
