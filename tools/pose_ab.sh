R=$PWD
for v in "" "B3D_HOIST=0" "B3D_WS2=0"; do
  echo "== $v"
  env $v python bench.py --model pose --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_median'), {k:round(v['us_per_step'],1) for k,v in d.get('kernels',{}).items()} if 'kernels' in d else '')"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pose_prof -o k -- python3 $R/bench.py --model pose --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > /dev/null 2>&1
