#!/usr/bin/env python3
"""Per-kernel averages of the derived on-chip metrics collected by tools/profile_onchip.sh (VALUBusy, SALUBusy,
VALUUtilization, LDSBankConflict, MemUnitBusy, MemUnitStalled, WriteUnitStalled ...; definitions: rocprofv3 --list-avail),
over each kernel's last full-size launches (the child steps), as tools/summarize_profiles.py selects them.
usage: summarize_onchip.py <gpurun_out/prof_TAG> <out.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

from summarize_profiles import short_name, steady

src, dst = sys.argv[1], sys.argv[2]
per = defaultdict(lambda: defaultdict(list))          # kernel -> metric -> [(grid, value)] in dispatch order
for path in sorted(glob.glob(os.path.join(src, "onchip_*", "**", "*counter_collection.csv"), recursive=True)):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        name = short_name(r["Kernel_Name"])
        if name:
            per[name][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
metrics = sorted({m for c in per.values() for m in c})
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches"] + metrics)
    for name in sorted(per):
        vals, n = [], 0
        for m in metrics:
            ls = per[name].get(m)
            if not ls:
                vals.append("")
                continue
            st = steady(ls, lambda x: x[0])
            n = max(n, len(st))
            vals.append(round(sum(v for _, v in st) / len(st), 2))
        w.writerow([name, n] + vals)
print(open(dst).read())
