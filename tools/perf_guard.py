"""Step-time regression guard: a fresh bench.py line against the last tracked profiles/*_bench_default.json.

    python bench.py > /tmp/line.json && python tools/perf_guard.py /tmp/line.json [--baseline FILE] [--tolerance 0.03]

Compares ms_per_step of the primary workload and of every secondary workload both lines carry (camera+LiDAR+radar with
precomputed encoder outputs, camera+LiDAR, PoseGNN ...); exits 1 when one of them is more than `tolerance` slower.
The round-2 line lost 7 % on the PoseGNN step (0.920 -> 0.988 ms: the slab reduction's scalar bias path) without
anything failing; this is the check that would have."""
import argparse, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_line(path):
    lines = [l for l in open(path).read().splitlines() if l.strip().startswith("{")]
    return json.loads(lines[-1])


def step_times(line):
    out = {"primary: " + line["config"]["workload"][:60]: float(line["ms_per_step"])}
    for name, sec in (line.get("secondary") or {}).items():
        if isinstance(sec, dict) and "ms_per_step" in sec:
            out["secondary." + name] = float(sec["ms_per_step"])
    return out


def default_baseline():
    def key(p):
        m = re.match(r"r(\d+)_([a-z]+)_", os.path.basename(p))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")), key=key)
    if not files:
        sys.exit("no profiles/r*_bench_default.json to compare with")
    return files[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("line")
    ap.add_argument("--baseline")
    ap.add_argument("--tolerance", type=float, default=0.03)
    a = ap.parse_args()
    base_path = a.baseline or default_baseline()
    new, old = step_times(last_line(a.line)), step_times(last_line(base_path))
    print(f"baseline: {os.path.relpath(base_path, ROOT)}")
    bad = 0
    for k in new:
        match = k if k in old else next((o for o in old if o.split(":")[0] == k.split(":")[0] and k.startswith("primary")), None)
        if match is None:
            print(f"  {k:75s} {new[k]:8.4f} ms   (not in the baseline)")
            continue
        r = new[k] / old[match] - 1.0
        flag = "REGRESSION" if r > a.tolerance else ""
        bad += bool(flag)
        print(f"  {k:75s} {old[match]:8.4f} -> {new[k]:8.4f} ms  {100 * r:+6.1f} %  {flag}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
