for k in 1 2; do for W in 8 4 16; do python bench.py --cpu-sweeps 0 --lanes-per-row $W 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); f=d['fp32']
print('W=$W f64 %.2f kernel %.2f   f32 %.2f kernel %.2f' % (1e3*d['ms_per_step'], d['roofline']['launch_us'], 1e3*f['ms_per_step'], f['roofline']['launch_us']))"; done; done
