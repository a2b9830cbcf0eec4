# A/B of environment switches in one GPU session: bash tools/ab_env2.sh "A=1 B=0" "A=0 B=0" ...   (two rounds, interleaved)
for round in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --steps ${AB_STEPS:-40} --warmup 5 --no-cpu-baseline --no-secondary ${AB_ARGS} > /tmp/ab_env2.out 2>/tmp/ab_env2.err
  tail -1 /tmp/ab_env2.out | python -c "
import json,sys,os
try:
    d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['ms_per_step_median'], d['replay_vs_eager_loss'].get('equal'), 'prologue', d['host_prologue_ms_median'], 'enqueue', d['host_enqueue_ms_per_step'])
except Exception as e:
    print('$cfg FAILED'); os.system('tail -5 /tmp/ab_env2.err')"
done
done
