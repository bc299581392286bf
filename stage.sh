for st in ${STAGES:-1 5 2 3 4 0}; do
  echo -n "stop=$st "
  ERM_TINY_STOP=1 ERM_PASS_STOP=$st python bench.py --steps 100 --warmup 10 --cpu-sweeps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'])"
done
