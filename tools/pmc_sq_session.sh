#!/bin/bash
# SQ counter passes of the default bench workload (eager launches): bash tools/pmc_sq_session.sh TAG
# Summaries: python tools/pmc_summary.py gpurun_out/TAG_sq1 ... ; python tools/pmc_ratios.py gpurun_out/TAG_sq*
TAG=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-graph $@"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/${TAG}_sq1 -o a -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/${TAG}_sq1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/${TAG}_sq2 -o b -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/${TAG}_sq2.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/${TAG}_sq3 -o c -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/${TAG}_sq3.err
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_sq1 > gpurun_out/${TAG}_pmc_mfma_busy.txt 2>&1
python tools/pmc_ratios.py gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_sq3 > gpurun_out/${TAG}_pmc_ratios.txt 2>&1
head -40 gpurun_out/${TAG}_pmc_mfma_busy.txt | cut -c1-260
