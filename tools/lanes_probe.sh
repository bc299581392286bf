# lanes-per-row probe (GPU box, repo root): the headline workload at W = 8 / 4 / 2 / 16 lanes per subject in the row-sum phase, twice
mkdir -p gpurun_out/r4
for k in 1 2; do for w in 8 4 2 16; do python bench.py --steps 300 --warmup 30 --no-two-chains --cpu-sweeps 0 --lanes-per-row $w 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); f=d.get('fp32') or {}
print('W=$w f64 %.2f kernel %.2f   f32 %.2f' % (1e3*d['ms_per_step'], d['roofline']['launch_us'], 1e3*f['ms_per_step']))"; done; done
