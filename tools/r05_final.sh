#!/bin/bash
# final validation of a tree whose library already has its PMC / kernel-trace session: smoke, all GPU tests, the default bench line, the one-rank collective line
TAG=${1:-r05_ad}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputest.log 2>&1; tail -3 gpurun_out/${TAG}_gputest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; tail -c 300 gpurun_out/${TAG}_bench_default.json; echo
python bench.py --steps 20 --warmup 5 --force-collective --no-secondary --no-cpu-baseline > gpurun_out/${TAG}_bench_fc.json 2> /dev/null; tail -c 150 gpurun_out/${TAG}_bench_fc.json
