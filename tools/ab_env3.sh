# A/B of environment switches with the kernel-family table: bash tools/ab_env3.sh "A=1" "A=0" ...   (two rounds, interleaved)
for round in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --steps ${AB_STEPS:-40} --warmup 5 --no-cpu-baseline --no-secondary ${AB_ARGS} > /tmp/ab_env3.out 2>/tmp/ab_env3.err
  tail -1 /tmp/ab_env3.out | python -c "
import json,sys,os
try:
    d=json.loads(sys.stdin.read()); k=d.get('kernels_instrumented_warmup') or d['kernels']
    print('$cfg', d['ms_per_step'], d['ms_per_step_median'], d['replay_vs_eager_loss'].get('equal'), {f:round(v['us_per_step'],1) for f,v in k.items() if f in ('wgrad_edge','wgrad_other','mp_edge_fwd','mp_edge_bwd')})
except Exception as e:
    print('$cfg FAILED'); os.system('tail -5 /tmp/ab_env3.err')"
done
done
