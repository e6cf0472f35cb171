#!/usr/bin/env python3
"""Condenses the three rocprofv3 runs of tools/profile_round.sh into the small files kept under profiles/:
kernel_stats.csv (the --stats table, engine kernels only), pmc_FETCH_SIZE.csv / pmc_WRITE_SIZE.csv (one row
per engine-kernel dispatch) and traffic.json (HBM bytes per launch and per kernel, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for wide streaming reads on gfx950; both counters are in KiB)."""
import csv
import glob
import json
import os
import sys

SHORT = {"scan_part_kernel": "scan_part", "repart_kernel": "repart", "seg_probe_kernel": "seg_probe",
         "seg_count_kernel": "seg_count", "seg_insert_kernel": "seg_insert", "seg_exact_probe_kernel": "seg_exact_probe",
         "seg_exact_insert_kernel": "seg_exact_insert", "probe_direct_kernel": "probe_direct",
         "insert_direct_kernel": "insert_direct", "count_insert_kernel": "count_insert", "count_emit_kernel": "count_emit"}


def short_name(kernel):
    for key, val in SHORT.items():
        if "dk::" + key in kernel:
            return val
    return None


def find(root, pattern):
    hits = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {root}")
    return hits[0]


def bench_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    return None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    # 1. kernel stats
    rows = list(csv.DictReader(open(find(os.path.join(src, "stats"), "*kernel_stats.csv"))))
    keep = [r for r in rows if "dk::" in r["Name"]]
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        w.writerows(keep)
    line = bench_line(os.path.join(src, "bench_under_stats.log"))
    if line:
        json.dump(line, open(os.path.join(dst, "bench_line_under_profiler.json"), "w"))
    # 2. PMC passes
    per = {}
    for tag, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = find(os.path.join(src, tag), "*counter_collection.csv")
        out_rows = []
        for r in csv.DictReader(open(path)):
            name = short_name(r["Kernel_Name"])
            if name is None or r["Counter_Name"] != counter:
                continue
            out_rows.append({"Dispatch_Id": r["Dispatch_Id"], "Kernel": name, "Grid_Size": r["Grid_Size"],
                             "Workgroup_Size": r["Workgroup_Size"], "LDS_Block_Size": r["LDS_Block_Size"],
                             "VGPR_Count": r["VGPR_Count"], "Counter_Name": counter, "Counter_Value_KiB": r["Counter_Value"]})
            per.setdefault(name, {}).setdefault(counter, []).append(float(r["Counter_Value"]))
        with open(os.path.join(dst, f"pmc_{counter}.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(out_rows[0].keys()))
            w.writeheader()
            w.writerows(out_rows)
    cfg = (line or {}).get("config", {})
    traffic = {"note": "HBM bytes per launch from rocprofv3 PMC (separate --pmc passes, tools/profile_round.sh). FETCH_SIZE "
                       "and WRITE_SIZE are in KiB; FETCH_SIZE is doubled for wide coalesced streaming reads on gfx950 "
                       "(MI355X_MICROARCH.md, HBM section). The child-pass kernels are averaged over the launches of the "
                       "largest grid size seen (the parent inserts launch the partition kernels too, with the same shapes).",
               "reads": cfg.get("reads_per_gpu"), "log2_bits": cfg.get("filter_log2_bits"), "kernels": {}}
    for name, c in per.items():
        f, wv = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
        if not f or not wv:
            continue
        fm, wm = sum(f) / len(f), sum(wv) / len(wv)
        traffic["kernels"][name] = {"fetch_bytes_corrected": fm * 1024 * 2, "write_bytes": wm * 1024,
                                    "hbm_bytes_per_launch": fm * 1024 * 2 + wm * 1024,
                                    "fetch_size_raw_kib": fm, "write_size_raw_kib": wm, "launches_seen": min(len(f), len(wv))}
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 2) for k, v in traffic["kernels"].items()}))
    for r in keep[:8]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])


if __name__ == "__main__":
    main()
