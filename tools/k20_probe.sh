for k in 1 2 3 4; do for L in "$@"; do if [ "$L" = "-" ]; then unset ERM_LIB_PATH; else export ERM_LIB_PATH=$PWD/$L; fi; python bench.py --steps 20 --warmup 5 --cpu-sweeps 0 --no-two-chains --no-fp32 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-45s K=20 f64 %.2f us (cold %.2f) live %.2f dev %.2f' % ('$L', 1e3*d['ms_per_step'], 1e3*d['ms_per_step_cold'], d['roofline']['launch_us'], 1e3*d['device_ms_per_step']))"; done; done
