import csv,glob,collections,sys
tag=sys.argv[1]; pats=sys.argv[2:]
f=glob.glob(f"gpurun_out/{tag}_kt/**/*kernel_trace.csv", recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if any(s in k for s in pats): print(f"{sum(v)/len(v):8.1f} us avg x{len(v):4d}  {k[:150]}")
