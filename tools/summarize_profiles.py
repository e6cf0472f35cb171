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
         "seg_probe_walk_kernel": "seg_probe", "seg_insert_walk_kernel": "seg_insert", "kmers_tile_kernel": "kmers",
         "seg_count_kernel": "seg_count", "seg_insert_kernel": "seg_insert", "seg_exact_probe_kernel": "seg_exact_probe",
         "seg_exact_insert_kernel": "seg_exact_insert", "probe_direct_kernel": "probe_direct",
         "insert_direct_kernel": "insert_direct", "count_insert_kernel": "count_insert", "count_emit_kernel": "count_emit"}


def short_name(kernel):
    for key, val in SHORT.items():
        if "dk::" + key + "<" in kernel or "dk::" + key + "(" in kernel:
            return val
    return None


LAST = 7      # the timed steps (and their warm-up) are the last launches of every kernel: bench.py --steps 5 --warmup 2


PER_STEP = {}     # kernel -> launches per step (slab-wise stages: one launch per level-2 slab), set from the bench line


def steady(ls, grid_of, name=None):
    """the launches of a kernel's last LAST steps in its steady shape: of its final 3 * LAST * per_step launches, those with
    the grid of the larger of the last two (a dk_probe step launches the membership kernel twice -- on 64 sampled segments,
    then on all); a slab-wise stage launches its kernel once per slab, PER_STEP[name] times per step"""
    per = PER_STEP.get(name, 1)
    top = max(int(grid_of(x) or 0) for x in ls[-2:])
    return [x for x in ls[-3 * LAST * per:] if int(grid_of(x) or 0) == top][-LAST * per:]


def trace_summary(src, dst):
    """per engine kernel: average duration of its last LAST launches (the child steps; the parent build, which runs the
    partition kernels on other shapes, comes first) and of all launches, from the kernel trace itself"""
    path = find(os.path.join(src, "stats"), "*kernel_trace.csv")
    by = {}
    for r in csv.DictReader(open(path)):
        name = short_name(r["Kernel_Name"])
        if name is None:
            continue
        by.setdefault(name, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                        r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", ""))))
    rows = []
    for name, ls in sorted(by.items()):
        ls.sort()
        last = steady(ls, lambda x: x[2], name)
        per = PER_STEP.get(name, 1)
        rows.append({"Kernel": name, "Calls": len(ls), "AverageNs_all": round(sum(x[1] for x in ls) / len(ls)),
                     "Last_launches": len(last), "AverageNs_last": round(sum(x[1] for x in last) / len(last)),
                     "Launches_per_step": per, "Ns_per_step_last": round(sum(x[1] for x in last) / (len(last) / per)),
                     "Grid_Size_last": last[-1][2], "Workgroup_Size_last": last[-1][3]})
    with open(os.path.join(dst, "kernel_trace_summary.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    return rows


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
        slabs = int(line.get("config", {}).get("level2_slabs") or 1)
        if line.get("config", {}).get("name") == "wgs" and slabs > 1:
            for name in ("repart", "seg_probe", "seg_exact_probe"):
                PER_STEP[name] = slabs
    lines = [l for l in open(os.path.join(src, "bench_under_stats.log")) if l.startswith("{")]
    if len(lines) > 1:                       # drivers that print one line per configuration (tools/kmers_bench.py)
        open(os.path.join(dst, "bench_lines_under_profiler.jsonl"), "w").writelines(lines)
    trows = trace_summary(src, dst)
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
            per.setdefault(name, {}).setdefault(counter, []).append((int(r["Grid_Size"]), float(r["Counter_Value"])))
        with open(os.path.join(dst, f"pmc_{counter}.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(out_rows[0].keys()))
            w.writeheader()
            w.writerows(out_rows)
    cfg = (line or {}).get("config", {})
    traffic = {"note": "HBM bytes per launch from rocprofv3 PMC (separate --pmc passes, tools/profile_round.sh). FETCH_SIZE "
                       "and WRITE_SIZE are in KiB; FETCH_SIZE is doubled for wide coalesced streaming reads on gfx950 "
                       "(MI355X_MICROARCH.md, HBM section). Every kernel is averaged over its last %d launches of full size = the child "
                       "steps (the parent inserts, which launch the partition kernels on other shapes, come first; the sampling "
                       "launch of the membership kernel on 64 segments is left out)." % LAST,
               "workload": cfg.get("name"), "reads": cfg.get("reads_per_sample"), "reads_per_step": cfg.get("reads_per_step"),
               "log2_bits": cfg.get("filter_log2_bits"), "kernels": {}}
    for name, c in per.items():
        if not c.get("FETCH_SIZE") or not c.get("WRITE_SIZE"):
            continue
        f = [v for _, v in steady(c["FETCH_SIZE"], lambda x: x[0], name)]
        wv = [v for _, v in steady(c["WRITE_SIZE"], lambda x: x[0], name)]
        fm, wm = sum(f) / len(f), sum(wv) / len(wv)
        per = PER_STEP.get(name, 1)
        traffic["kernels"][name] = {"fetch_bytes_corrected": fm * 1024 * 2, "write_bytes": wm * 1024,
                                    "hbm_bytes_per_launch": fm * 1024 * 2 + wm * 1024,
                                    "launches_per_step": per, "hbm_bytes_per_step": (fm * 1024 * 2 + wm * 1024) * per,
                                    "fetch_size_raw_kib": fm, "write_size_raw_kib": wm, "launches_seen": min(len(f), len(wv))}
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_step"] / 1e9, 2) for k, v in traffic["kernels"].items()}))
    for r in trows:
        print(r)


if __name__ == "__main__":
    main()
