#!/bin/bash
# kernel trace of the PoseGNN training step under the library variant $2: per-kernel averages -> stdout
TAG=$1; VAR=$2
R=$PWD
cd /tmp && export TMPDIR=/tmp
[ -n "$VAR" ] && export B3D_LIB=$VAR
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_kt -o k -- python3 $R/bench.py --model pose --steps 20 --warmup 3 --no-secondary --no-cpu-baseline --ramp-ms 0 > $R/gpurun_out/${TAG}_kt.json 2> $R/gpurun_out/${TAG}_kt.err
cd $R
python tools/kt_names.py $TAG wgrad knn_tile reduce adam | head -8
