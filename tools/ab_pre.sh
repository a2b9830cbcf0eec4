for round in 1 2 3; do
for v in 0 1; do
  B3D_SKIP_PRE=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('skip_pre=$v', d['ms_per_step'], d['ms_per_step_median'], d['host_enqueue_ms_per_step'], d['host_graph_launch_ms_median'], d['host_prologue_ms_median'])"
done
done
