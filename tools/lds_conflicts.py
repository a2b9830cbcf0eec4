"""LDS bank-conflict ratio (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) per kernel from a rocprofv3 --pmc run:
python tools/lds_conflicts.py DIR [substring ...]"""
import collections, csv, glob, sys
agg = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("b3d::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        if len(sys.argv) > 2 and not any(s in n for s in sys.argv[2:]):
            continue
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"]):
    print("%.3f  %12.0f  %s" % (c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), c["SQ_LDS_IDX_ACTIVE"], n))
