"""Per-kernel means of rocprofv3 --pmc counters: python tools/pmc_summary.py DIR [DIR...]"""
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('b3d::', '').replace('MPDims<48, 32, 0, 96, 64, 96, 64, 96, 64>', 'P').split('(')[0][:60]
            agg[n][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[n][r['Counter_Name']] += 1
            if 'Start_Timestamp' in r and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
names = sorted(agg, key=lambda n: -sum(agg[n].values()))
for n in names[:40]:
    c = agg[n]
    parts = [f"{k}={c[k] / cnt[n][k]:.4g}" for k in sorted(c)]
    extra = ""
    if dur[n] and 'GRBM_GUI_ACTIVE' in c:
        us = sum(dur[n]) / len(dur[n]) / 1e3
        act = c['GRBM_GUI_ACTIVE'] / cnt[n]['GRBM_GUI_ACTIVE'] / 8          # per XCD
        # GRBM_GUI_ACTIVE covers the whole dispatch (command processor, ramp-up, drain), not only the kernel's Start..End timestamps:
        # active / duration is NOT a clock -- for launches under ~50 us it exceeded the part's 2.4 GHz in round 5's tables (2.8-3.15
        # "GHz").  Round 6: it is printed as what it is, the dispatch overhead it implies at 2.4 GHz is shown, and the matrix-pipe
        # figure is given against BOTH denominators: the GRBM-active cycles (a lower bound of the busy share of the kernel proper)
        # and duration x 2.4 GHz (a lower bound whenever the chip clocks below 2.4 GHz, which it does under MFMA load).
        over_us = max(0.0, act / 2.4e9 * 1e6 - us)
        extra = f"  | avg {us:.1f} us; GRBM-active cycles / duration = {act / (us * 1e-6) / 1e9:.2f} G/s (>= {over_us:.1f} us of dispatch outside the timestamps at 2.4 GHz)"
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            # MFMA busy cycles are summed over all SIMDs of the chip (256 CU x 4)
            busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / cnt[n]['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024
            extra += (f", MFMA pipe busy {100 * busy / act:.0f}% of GRBM-active cycles, {100 * busy / (us * 1e-6 * 2.4e9):.0f}% of duration x 2.4 GHz"
                      + (" [short launch: soft numbers]" if us < 50 else ""))
    print(f"{n} (x{max(cnt[n].values())})\n    " + "  ".join(parts) + extra)
