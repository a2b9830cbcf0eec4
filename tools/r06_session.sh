#!/bin/bash
# Round-6 session (same passes as round 5) on the GPU box (from the repo root): GPU tests, the default bench line, rocprofv3 kernel trace + stats of the
# default command, PMC traffic passes (training + inference workloads), SQ ratio passes.  Outputs under gpurun_out/$1_*.
TAG=$1
R=$PWD
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputest.log 2>&1; tail -3 gpurun_out/${TAG}_gputest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; tail -c 200 gpurun_out/${TAG}_bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o k -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_prof_bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o f -- python3 $R/bench.py $ARGS --no-graph > /dev/null 2> $R/gpurun_out/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o w -- python3 $R/bench.py $ARGS --no-graph > /dev/null 2> $R/gpurun_out/${TAG}_pmc_write.err
IARGS="--mode infer --steps 1 --warmup 4 --no-cpu-baseline --no-graph"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_infer -o f -- python3 $R/bench.py $IARGS > /dev/null 2> $R/gpurun_out/${TAG}_pmc_fetch_infer.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_infer -o w -- python3 $R/bench.py $IARGS > /dev/null 2> $R/gpurun_out/${TAG}_pmc_write_infer.err
cd $R
python tools/pmc_traffic.py clr:frozen:knn1 gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_traffic.json gpurun_out/${TAG}_pmc_traffic.txt > /dev/null
python tools/pmc_traffic.py clr:infer:knn1 gpurun_out/${TAG}_pmc_fetch_infer gpurun_out/${TAG}_pmc_write_infer gpurun_out/${TAG}_traffic.json gpurun_out/${TAG}_pmc_traffic_infer.txt > /dev/null
python tools/rocprof_per_step.py gpurun_out/${TAG}_prof/k_kernel_stats.csv 0 80 > gpurun_out/${TAG}_last_step_summary.txt 2>&1
# the raw counter files are large: keep the summaries
rm -rf gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_fetch_infer gpurun_out/${TAG}_pmc_write_infer
bash tools/pmc_sq_session.sh ${TAG} > /dev/null 2>&1
rm -rf gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_sq3
ls gpurun_out | grep ${TAG}
# randomised parity sweeps on the same library
( echo "# python tools/fuzz_parity.py --cases 24 (seed 0), library $(sha256sum batch3dmot_amd/libb3d_hip.so | cut -c1-16)"; python tools/fuzz_parity.py --cases 24; echo; echo "# python tools/fuzz_parity.py --cases 30 --seed 7"; python tools/fuzz_parity.py --cases 30 --seed 7 ) > gpurun_out/${TAG}_fuzz_parity.txt 2>&1
tail -1 gpurun_out/${TAG}_fuzz_parity.txt
