# HIP's hardware-queue budget per process (GPU_MAX_HW_QUEUES, default 4): streams beyond it share a queue and serialize
for round in 1 2; do
for q in 4 3 5 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q', d['ms_per_step'], d['ms_per_step_median'], d['host_enqueue_ms_per_step'], d['host_graph_launch_ms_median'], d['host_prologue_ms_median'], d['replay_vs_eager_loss']['equal'])"
done
done
