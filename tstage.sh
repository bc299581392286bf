cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for st in 1 2 3 4 0; do ERM_NO_GRAPH=1 ERM_TINY_STOP=$st rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t$st -- python3 bench.py --steps 30 --warmup 3 --cpu-sweeps 0 --no-profile > gpurun_out/prof_t.log 2>&1; echo -n "stop=$st "; grep tiny gpurun_out/prof_t$st/*/*_kernel_stats.csv | cut -d, -f4,6,7; done
