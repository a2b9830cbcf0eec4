TAG=${1:-r05_t}
R=$PWD
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputest.log 2>&1; tail -3 gpurun_out/${TAG}_gputest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; tail -c 200 gpurun_out/${TAG}_bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o k -- python3 $R/bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_prof_bench.err
cd $R
python tools/rocprof_per_step.py gpurun_out/${TAG}_prof/k_kernel_stats.csv 0 90 > gpurun_out/${TAG}_last_step_summary.txt 2>&1
python bench.py --steps 20 --warmup 5 --force-collective --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_bench_fc.json 2> /dev/null; tail -c 150 gpurun_out/${TAG}_bench_fc.json
