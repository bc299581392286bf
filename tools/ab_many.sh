V=extendedrtirtmodeling.jl_amd/libertirt_t1024.so
for args in "--model crossqr" "--model mlirt" "--model latentqr" "--model cross" "--nsubj 500000 --nitem 100 --steps 200 --warmup 20" "--steps 20 --warmup 5" "--nsubj 1000 --nitem 15 --model mlirt"; do
  echo "== $args"
  bash tools/ab_bench.sh --no-fp32 $args -- - $V | head -4
done
