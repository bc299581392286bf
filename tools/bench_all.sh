#!/bin/bash
# Every bench line kept under profiles/ (run on the GPU box from the repo root, after tools/collect_profiles.sh):
# the default command (with its CPU baselines), the other samplers at the same size, the 500k x 100 configuration, the fp64 engine
# and the summary-trace mode.  Output: gpurun_out/<tag>/lines/<name>.json, one JSON line each.
TAG=${1:-round1}
OUT=gpurun_out/$TAG/lines; mkdir -p $OUT
run() { n=$1; shift; timeout -k 10 400 python bench.py "$@" 2>$OUT/$n.err | tail -1 > $OUT/$n.json; echo "$n $(grep -o '"ms_per_step": [0-9.]*' $OUT/$n.json)"; }
run line
for m in mlirt latentqr crossqr null cross latent; do run $m --model $m; done
run 500kx100 --nsubj 500000 --nitem 100 --steps 200 --warmup 20 --cpu-sweeps 0
run f64 --precision f64 --steps 200 --warmup 20 --cpu-sweeps 0
run summary --trace summary --cpu-sweeps 0
