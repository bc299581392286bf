# everything profiles/ holds for a round, collected on ONE box (GPU box, repo root): every bench line, the rocprofv3 kernel stats and counters of both engines,
# the VALU class mix, the stage budget.  Afterwards, here: copy gpurun_out/round4/lines/*.json to profiles/round4_bench_*.json, tools/summarize_profiles.py, tools/valu_weighted.py.
mkdir -p gpurun_out
bash tools/bench_all.sh round4 > gpurun_out/bench_all.log 2>&1
bash tools/collect_profiles.sh round4 f64 > gpurun_out/collect_f64.log 2>&1
bash tools/collect_profiles.sh round4 f32 > gpurun_out/collect_f32.log 2>&1
bash tools/valu_classes.sh round4 f64 > gpurun_out/valu_f64.log 2>&1
bash tools/valu_classes.sh round4 f32 > gpurun_out/valu_f32.log 2>&1
python3 tools/stage_budget.py --precisions f64 > gpurun_out/budget.log 2>&1
cat gpurun_out/bench_all.log
