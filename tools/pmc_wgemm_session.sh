#!/bin/bash
# SQ counter passes for ONE library variant (B3D_LIB=name or unset), wgemm_kernel rows only -> gpurun_out/$1_wgemm_pmc.txt
TAG=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --no-secondary --no-cpu-baseline --no-graph --ramp-ms 0"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_wg$i -o p -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/${TAG}_wg$i.err
done
cd $R
python - <<PY
import csv,glob,collections
tot=collections.defaultdict(float);cnt=collections.Counter()
for f in glob.glob("gpurun_out/${TAG}_wg*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgemm_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
with open("gpurun_out/${TAG}_wgemm_pmc.txt","w") as fh:
    for k in sorted(tot):
        line=f"{k:28s} {tot[k]/cnt[k]/1e6:10.2f} M per launch ({cnt[k]} launches)"
        print(line); fh.write(line+"\n")
PY
