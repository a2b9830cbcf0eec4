#!/bin/bash
# One GPU-box session: kernel-trace stats + the two PMC passes of the default bench workload; summaries land in
# gpurun_out/$1_*.  Usage (on the GPU box, from the repo root): bash tools/gpu_profile_session.sh TAG [bench args]
TAG=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o k -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof_bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o f -- python3 $R/bench.py $ARGS --no-graph > /dev/null 2> $R/gpurun_out/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o w -- python3 $R/bench.py $ARGS --no-graph > /dev/null 2> $R/gpurun_out/${TAG}_pmc_write.err
cd $R
# HBM bytes per launch per family -> gpurun_out/${TAG}_traffic.json (copy to profiles/traffic_pmc.json: bench.py reads it by workload key)
python tools/pmc_traffic.py clr:frozen:knn1 gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_traffic.json gpurun_out/${TAG}_pmc_traffic.txt > /dev/null
# the same two counter passes for the inference workload (BASELINE.json configs[4]; bench.py --mode infer, eager) -> key clr:infer:knn1
cd /tmp
IARGS="--mode infer --steps 1 --warmup 4 --no-cpu-baseline --no-graph"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_infer -o f -- python3 $R/bench.py $IARGS > /dev/null 2> $R/gpurun_out/${TAG}_pmc_fetch_infer.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_infer -o w -- python3 $R/bench.py $IARGS > /dev/null 2> $R/gpurun_out/${TAG}_pmc_write_infer.err
cd $R
python tools/pmc_traffic.py clr:infer:knn1 gpurun_out/${TAG}_pmc_fetch_infer gpurun_out/${TAG}_pmc_write_infer gpurun_out/${TAG}_traffic.json gpurun_out/${TAG}_pmc_traffic_infer.txt > /dev/null
ls gpurun_out/${TAG}_prof gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write | head -20
