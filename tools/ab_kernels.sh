#!/bin/bash
# Per-kernel A/B of library variants by rocprofv3 kernel stats (one short bench each, same box): bash tools/ab_kernels.sh "<grep pattern>" default v1 ...
PAT=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "default" ]; then unset B3D_LIB; else export B3D_LIB=$v; fi
  rm -rf /tmp/abk_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$v -o k -- python3 $R/bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > /tmp/abk_$v.json 2> /tmp/abk_$v.err
  echo "== $v"
  python3 $R/tools/rocprof_per_step.py /tmp/abk_$v/k_kernel_stats.csv 0 200 2>&1 | grep -E "$PAT" | cut -c1-170
done
