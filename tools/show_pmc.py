import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('b3d::', '').replace('MPDims<48, 32, 0, 96, 64, 96, 64, 96, 64>', 'P').split('(')[0][:48]
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': cnt[n] += 1
for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:12]:
    k = max(cnt[n], 1)
    print(n, f"(x{cnt[n]})")
    print("   " + "  ".join(f"{a}={v/k:.3g}" for a, v in sorted(c.items())))
