# A/B of library variants in one GPU session, one round: bash tools/ab_once.sh default abl1 abl2 ...
for v in "$@"; do
  if [ "$v" = "default" ]; then unset B3D_LIB; else export B3D_LIB=$v; fi
  python bench.py --steps ${AB_STEPS:-30} --warmup 5 --no-cpu-baseline --no-secondary ${AB_ARGS} > /tmp/ab_once.out 2>&1
  grep "^{" /tmp/ab_once.out | tail -1 | python -c "
import json,sys,os
try:
    d=json.loads(sys.stdin.read()); print(os.environ.get('B3D_LIB','default'), d['ms_per_step'], d['ms_per_step_median'], d['replay_vs_eager_loss'].get('equal'), {k:(round(v['us_per_step'],1)) for k,v in (d.get('kernels_instrumented_warmup') or d['kernels']).items() if k in ('mp_edge_fwd','mp_edge_bwd','wgrad_edge','att_fwd','att_bwd','mp_node_fwd','mp_node_bwd','other')})
except Exception as e:
    print(os.environ.get('B3D_LIB','default'), 'FAILED'); os.system('tail -5 /tmp/ab_once.out')"
done
