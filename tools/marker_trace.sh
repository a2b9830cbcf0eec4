#!/bin/bash
# roctx ranges of the kernel families in a rocprofv3 marker trace (b3d_prof_markers / B3D_ROCTX=1): two eager
# forward + backward passes of the camera+LiDAR+radar model.  Usage (GPU box, repo root): bash tools/marker_trace.sh TAG
TAG=${1:-markers}
R=$PWD
cd /tmp && export TMPDIR=/tmp
export B3D_ROCTX=1
rocprofv3 --marker-trace --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_markers -o m -- python3 $R/tools/run_clr_steps.py 2 > $R/gpurun_out/${TAG}_markers.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
fs = glob.glob("gpurun_out/${TAG}_markers/**/*marker_api_trace.csv", recursive=True)
out = open("gpurun_out/${TAG}_marker_ranges.txt", "w")
if not fs:
    out.write("no marker trace file was produced\n")
else:
    rows = list(csv.DictReader(open(fs[0])))
    cnt, dur = collections.Counter(), collections.Counter()
    for r in rows:
        name = r.get("Function") or r.get("Message") or r.get("Name") or "?"
        try:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        except Exception:
            d = 0
        cnt[name] += 1; dur[name] += d
    out.write("# roctx ranges emitted by libb3d_hip.so around its kernel-family launches (host-side ranges, eager step), rocprofv3 --marker-trace\n")
    out.write("# columns: ranges, total host time inside them (us), name\n")
    for n, c in cnt.most_common():
        out.write(f"{c:6d}  {dur[n] / 1e3:10.1f}  {n}\n")
out.close()
print(open("gpurun_out/${TAG}_marker_ranges.txt").read())
PY
