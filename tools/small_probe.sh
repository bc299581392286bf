# small-data A/B (GPU box): bash tools/small_probe.sh lib1.so lib2.so ...  ("-" = in-tree)
for k in 1 2 3; do for L in "$@"; do if [ "$L" = "-" ]; then unset ERM_LIB_PATH; else export ERM_LIB_PATH=$PWD/$L; fi
for m in mlirt rtirt; do python bench.py --model $m --nsubj 1000 --nitem 15 --steps 2000 --warmup 100 --cpu-sweeps 0 --no-two-chains 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); f=d.get('fp32') or {}
print('%-45s $m 1000x15 f64 %.2f us  f32 %.2f us' % ('$L', 1e3*d['ms_per_step'], 1e3*f['ms_per_step']))"; done; done; done
