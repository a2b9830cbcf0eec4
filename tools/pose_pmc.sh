R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pose_pmc_f -o f -- python3 $R/bench.py --model pose --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-graph > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pose_pmc_w -o w -- python3 $R/bench.py --model pose --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-graph > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py pose:na:knn1 gpurun_out/pose_pmc_f gpurun_out/pose_pmc_w gpurun_out/pose_traffic.json gpurun_out/pose_pmc_traffic.txt
head -12 gpurun_out/pose_pmc_traffic.txt | cut -c1-180
