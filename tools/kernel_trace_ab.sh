#!/bin/bash
# kernel trace of a few eager steps under the library variant $2 (B3D_LIB; "" = shipped); prints per-launch durations of kernels matching $3
TAG=$1; VAR=$2; PAT=$3
R=$PWD
cd /tmp && export TMPDIR=/tmp
[ -n "$VAR" ] && export B3D_LIB=$VAR
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_kt -o k -- python3 $R/bench.py --steps 6 --warmup 3 --no-secondary --no-cpu-baseline --ramp-ms 0 > $R/gpurun_out/${TAG}_kt.json 2> $R/gpurun_out/${TAG}_kt.err
cd $R
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/${TAG}_kt/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sel=[r for r in rows if "${PAT}" in r["Kernel_Name"]]
per=collections.defaultdict(list)
# launches of the last step: the pattern repeats; print the last 12 matching launches with grid sizes
out=open("gpurun_out/${TAG}_kt_sel.txt","w")
for r in sel[-14:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    line=f'{d:8.1f} us  grid {r["Grid_Size_X"]:>8s} wg {r["Workgroup_Size_X"]:>4s}  {r["Kernel_Name"][:90]}'
    print(line); out.write(line+"\n")
PY
tail -1 gpurun_out/${TAG}_kt.json | cut -c1-200
