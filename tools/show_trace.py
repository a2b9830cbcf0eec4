import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if 'graph_convert_count' in r['Kernel_Name']]
last = rows[idx[-1]:]
t0 = int(last[0]['Start_Timestamp'])
agg = {}
for r in last:
    n = r['Kernel_Name'].replace('b3d::', '').replace('MPDims<48, 32, 0, 96, 64, 96, 64, 96, 64>', 'P')
    n = n.split('(')[0][:70]
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += dur
    if len(sys.argv) > 2:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us dur {dur:8.1f} grid {r['Grid_Size_X']:>7s} wg {r['Workgroup_Size_X']:>4s} {n}")
tot = sum(v[1] for v in agg.values())
print(f"last step: kernel time sum {tot:.1f} us, span {(int(last[-1]['End_Timestamp'])-t0)/1e3:.1f} us")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t:9.1f} us  x{c:3d}  avg {t/c:7.1f}  {n}")
