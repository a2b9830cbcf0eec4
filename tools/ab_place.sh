for round in 1 2; do
for v in split split5 backward split3 start; do
  B3D_AHEAD_AT=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['ms_per_step_median'], d['replay_vs_eager_loss']['equal'])"
done
done
