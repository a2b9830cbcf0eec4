"""HBM bytes per launch of every kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes:
the TCC block cannot hold both), corrected as MI355X_MICROARCH.md prescribes for gfx950 (raw counters in KB; wide coalesced
reads are tallied at half their bytes): bytes = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE.

    python tools/pmc_traffic.py WORKLOAD_KEY FETCH_DIR WRITE_DIR [out.json] [table.txt]

Merges {WORKLOAD_KEY: {family: bytes per launch}} into out.json (default profiles/traffic_pmc.json, read by bench.py)."""
import collections, csv, glob, json, os, sys

FAMILY = [("mp_edge_fwd", "mp_edge_fwd"), ("mp_edge_bwd", "mp_edge_bwd"), ("edge_fwd_kernel", "mp_edge_fwd"), ("edge_bwd_kernel", "mp_edge_bwd"), ("mp_node_fwd", "mp_node_fwd"),
          ("node_bwd", "mp_node_bwd"), ("node_listsum", "mp_node_bwd"), ("node_gradproj", "mp_node_bwd"),
          ("wstream", "wgrad_edge"), ("wgemm", "wgrad_edge"), ("wgrad_kernel", "wgrad_other"), ("point_feat", "point_feat"), ("knn_", "knn_gat"),
          ("gat_", "knn_gat")]


def family(name):
    if "wide_linear_kernel" in name:
        # forward layers carry ReLU/bias template flags <..., true, true, ...>; the transposed (data-gradient) ones <false, false>
        return "att_bwd" if "false, false" in name else "att_fwd"
    for key, fam in FAMILY:
        if key in name:
            return fam
    return None


def read(d, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return tot, cnt


def main():
    key, fdir, wdir = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_pmc.json")
    table = sys.argv[5] if len(sys.argv) > 5 else None
    ft, fc = read(fdir, "FETCH_SIZE")
    wt, wc = read(wdir, "WRITE_SIZE")
    fam_b, fam_n = collections.defaultdict(float), collections.Counter()
    lines = []
    for k in sorted(set(ft) | set(wt), key=lambda k: -(2 * ft.get(k, 0) + wt.get(k, 0))):
        n = max(fc.get(k, 0), wc.get(k, 0))
        if n == 0:
            continue
        per = 1024.0 * (2.0 * ft.get(k, 0.0) / max(fc.get(k, 1), 1) + wt.get(k, 0.0) / max(wc.get(k, 1), 1))
        lines.append(f"{per / 1e6:10.2f} MB/launch  x{n:5d}  FETCH_SIZE {ft.get(k, 0) / max(fc.get(k, 1), 1):12.1f} KB  WRITE_SIZE {wt.get(k, 0) / max(wc.get(k, 1), 1):12.1f} KB  {k[:120]}")
        fam = family(k)
        if fam:
            fam_b[fam] += per * n
            fam_n[fam] += n
    res = {f: round(fam_b[f] / fam_n[f]) for f in fam_b}
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[key] = res
    # which library build and day the counters belong to (bench.py prints it beside roofline.traffic)
    import datetime, hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        sha = hashlib.sha256(open(os.path.join(root, "batch3dmot_amd", "libb3d_hip.so"), "rb").read()).hexdigest()[:16]
    except OSError:
        sha = None
    data["_source"] = {"lib_sha16": sha, "date": datetime.date.today().isoformat(), "tool": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"}
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    text = "\n".join(lines[:60])
    if table:
        open(table, "w").write(f"# {key}: corrected HBM bytes per launch = 1024 * (2 * FETCH_SIZE + WRITE_SIZE), rocprofv3 --pmc, separate passes\n" + text + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
