"""Per-kernel averages of the counters of one or more rocprofv3 --pmc output directories:
python tools/pmc_kernel_table.py SUBSTRING DIR [DIR ...]"""
import collections, csv, glob, os, sys
sub, dirs = sys.argv[1], sys.argv[2:]
tot, cnt = collections.defaultdict(float), collections.Counter()
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                k = (r["Kernel_Name"][:70], r["Counter_Name"])
                tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for (kn, c) in sorted(tot):
    print(f"{kn:70s} {c:28s} {tot[(kn, c)] / cnt[(kn, c)]:16.1f}  (x{cnt[(kn, c)]})")
