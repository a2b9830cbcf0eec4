"""Per-step kernel table from a rocprofv3 --kernel-trace --stats CSV: kernels grouped into the library's own and the
others (PyTorch / MIOpen / rocBLAS); one-off kernels (MIOpen's find pass on a fresh box) are listed apart.
usage: python tools/rocprof_per_step.py <kernel_stats.csv> <steps incl. warm-up> [top]"""
import csv
import sys


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rows = list(csv.DictReader(open(path)))
    if steps <= 0:                      # one graph build per executed step: count the steps from it
        steps = max(int(r["Calls"]) for r in rows if "graph_convert_count" in r["Name"])
        print(f"(steps executed, from the graph-build launches: {steps})")
    own, other, oneoff = [], [], []
    for r in rows:
        name, calls, total = r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3
        per = calls / steps
        rec = (total / steps, per, total / calls, name[:110])
        if "b3d" in name:
            own.append(rec)
        elif abs(per - round(per)) > 0.05 or per < 0.9:
            oneoff.append((total, calls, total / calls, name[:110]))
        else:
            other.append(rec)
    for title, lst in (("library kernels", own), ("other kernels (every step)", other)):
        lst.sort(reverse=True)
        print(f"== {title}: {sum(x[0] for x in lst):.1f} us/step, {sum(x[1] for x in lst):.0f} launches/step")
        for us, per, avg, name in lst[:top]:
            print(f"{us:9.1f} us/step  x{per:6.1f}  avg {avg:8.1f} us  {name}")
    oneoff.sort(reverse=True)
    print(f"== not per-step (warm-up / MIOpen find): {sum(x[0] for x in oneoff) / 1e3:.1f} ms in total")
    for total, calls, avg, name in oneoff[:10]:
        print(f"{total / 1e3:9.1f} ms  x{calls:6d}  avg {avg:8.1f} us  {name}")


if __name__ == "__main__":
    main()
