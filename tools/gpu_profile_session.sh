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
ls gpurun_out/${TAG}_prof gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write | head -20
