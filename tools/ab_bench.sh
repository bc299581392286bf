# A/B of two library builds on ONE box: alternating runs of the default bench (fp64 only)
OLD=$PWD/extendedrtirtmodeling.jl_amd/libertirt_old.so
for k in 1 2 3; do
  ERM_LIB_PATH=$OLD python bench.py --cpu-sweeps 0 --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('old', d['ms_per_step'], d['roofline']['launch_us'])"
  python bench.py --cpu-sweeps 0 --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new', d['ms_per_step'], d['roofline']['launch_us'])"
done
