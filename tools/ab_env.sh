# A/B of one environment switch in one GPU session: bash tools/ab_env.sh VAR value_a value_b [bench args]
VAR=$1; A=$2; B=$3; shift 3
for round in 1 2 3; do
for v in $A $B; do
  env $VAR=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels_instrumented_warmup') or d['kernels']
print('$VAR=$v', d['ms_per_step'], d['ms_per_step_median'], d['replay_vs_eager_loss']['equal'], {n:round(v['us_per_step'],1) for n,v in k.items() if n in ('point_feat','other','fc_heads')})"
done
done
