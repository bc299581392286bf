# chains-per-GPU experiment (one-GPU box): K chains of the C-ABI farm on device 0 under different launch geometries; tools/cpg_report.py prints the table
mkdir -p gpurun_out/r4 && export ERM_BENCH_REHEARSE=1
C="--steps 300 --warmup 30 --no-configs4 --no-self-check --no-fp32 --cpu-sweeps 0"
run() { tag=$1; shift; python bench.py $C "$@" > gpurun_out/r4/cpg_$tag.json 2> gpurun_out/r4/cpg_$tag.err || { echo "FAILED $tag"; tail -3 gpurun_out/r4/cpg_$tag.err; }; }
for m in null latent mlirt; do
run ${m}_2_default --gpus 2 --model $m
run ${m}_2_b512g512 --gpus 2 --model $m --block-threads 512 --grid-blocks 512
done
for sz in "20000 50" "50000 30" "200000 50" "100000 100" "100000 20"; do set -- $sz
run rtirt_$1x$2_2_default --gpus 2 --nsubj $1 --nitem $2
run rtirt_$1x$2_2_b512g512 --gpus 2 --nsubj $1 --nitem $2 --block-threads 512 --grid-blocks 512
done
run rtirt_500000x100_2_default --gpus 2 --nsubj 500000 --nitem 100 --steps 60 --warmup 10 --trace summary
run rtirt_500000x100_2_b512g1024 --gpus 2 --nsubj 500000 --nitem 100 --steps 60 --warmup 10 --trace summary --block-threads 512 --grid-blocks 1024
run rtirt_500000x100_2_b512g1536 --gpus 2 --nsubj 500000 --nitem 100 --steps 60 --warmup 10 --trace summary --block-threads 512 --grid-blocks 1536
echo done
