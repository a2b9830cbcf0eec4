#!/bin/bash
# standalone ResNetAE.encode (3,000 crops) under the kernel trace: per-phase kernel durations without the other encoder streams
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$1_kt -o k -- python3 $R/tools/bench_resnet_encode.py 3000 > $R/gpurun_out/$1_resnet.txt 2>&1
cd $R
tail -3 gpurun_out/$1_resnet.txt
python tools/kt_names.py $1 resnet_
