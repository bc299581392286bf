#!/bin/bash
# Every bench line kept under profiles/ (run on the GPU box from the repo root): the default command (fp64 headline + nested fp32, with its CPU
# baselines), the driver's short command, the other samplers at the same size, the 500k x 100 configuration and the summary-trace mode.
# Output: gpurun_out/<tag>/lines/<name>.json, one JSON line each.
TAG=${1:-round2}
OUT=gpurun_out/$TAG/lines; mkdir -p $OUT
run() { n=$1; shift; timeout -k 10 500 python bench.py "$@" 2>$OUT/$n.err | tail -1 > $OUT/$n.json; echo "$n $(grep -o '"ms_per_step": [0-9.]*' $OUT/$n.json | head -2 | tr '\n' ' ')"; }
run line
run driver20 --steps 20 --warmup 5 --cpu-sweeps 0
for m in mlirt latentqr crossqr null cross latent; do run $m --model $m --cpu-sweeps 0; done
run 500kx100 --nsubj 500000 --nitem 100 --steps 200 --warmup 20 --cpu-sweeps 0
run summary --trace summary --cpu-sweeps 0
# the N > 1 path rehearsed on this one-GPU box: two chains of the C-ABI farm on device 0, the reduction through the library's RCCL communicator (one rank)
ERM_BENCH_REHEARSE=1 run farm2_rehearsal --gpus 2 --steps 200 --warmup 20 --cpu-sweeps 0
