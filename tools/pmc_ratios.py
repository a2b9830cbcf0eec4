"""Per-kernel ratios from rocprofv3 --pmc directories (tools/pmc_kernel_table.py collects the same files):
MFMA busy cycles per MFMA instruction, share of wave-cycles in s_waitcnt, LDS conflict share.
python tools/pmc_ratios.py DIR [DIR ...]"""
import collections, csv, glob, os, sys
tot, cnt = collections.defaultdict(float), collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:90], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
names = sorted({k[0] for k in tot})
rows = []
for n in names:
    g = lambda c: tot.get((n, c), 0.0) / max(cnt.get((n, c), 1), 1)
    wc = g("SQ_WAVE_CYCLES")
    if wc < 1e6:
        continue
    mf = g("SQ_INSTS_MFMA")
    rows.append((wc, n, g("SQ_VALU_MFMA_BUSY_CYCLES") / mf if mf else 0.0, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc,
                 g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0), mf, g("SQ_INSTS_VALU") - mf))
for wc, n, cpm, wa, wl, lc, mf, va in sorted(rows, reverse=True):
    print(f"{wc / 1e6:8.1f}M wave-cyc  mfma cyc/inst {cpm:5.1f}  wait {wa:5.2f}  wait_lds {wl:5.2f}  lds_conflict {lc:5.2f}  mfma {mf / 1e6:6.2f}M valu {va / 1e6:6.2f}M  {n}")
