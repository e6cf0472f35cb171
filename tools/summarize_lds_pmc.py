#!/usr/bin/env python3
"""Per-kernel LDS occupancy from one rocprofv3 --pmc pass (SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT,
SQ_LDS_ADDR_CONFLICT, SQ_LDS_ATOMIC_RETURN, SQ_INSTS_LDS, GRBM_GUI_ACTIVE):
  lds_util_pct      = 100 * SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 * 256)   share of the kernel's cycles in which a CU's
                      LDS array is busy (both counters arrive summed over their instances: 256 CUs, 8 XCDs)
  bank_conflict_pct = 100 * SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
usage: summarize_lds_pmc.py <counter_collection.csv> <out.csv>"""
import csv
import sys
from collections import defaultdict

from summarize_profiles import short_name

rows = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = short_name(r["Kernel_Name"])
    if name:
        rows[(name, r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
per = defaultdict(lambda: defaultdict(list))
for (name, _), c in rows.items():
    for k, v in c.items():
        per[name][k].append(sum(v))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "lds_util_pct", "bank_conflict_pct_of_lds_active", "addr_conflict_pct_of_lds_active",
                "atomic_return_pct_of_lds_active", "lds_insts_per_launch", "kernel_cycles_per_launch"])
    for name, c in per.items():
        n = len(c["GRBM_GUI_ACTIVE"])
        avg = {k: sum(v) / len(v) for k, v in c.items()}
        idx = avg.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0
        w.writerow([name, n, round(100 * idx / (avg["GRBM_GUI_ACTIVE"] / 8 * 256), 1),
                    round(100 * avg.get("SQ_LDS_BANK_CONFLICT", 0) / idx, 1), round(100 * avg.get("SQ_LDS_ADDR_CONFLICT", 0) / idx, 1),
                    round(100 * avg.get("SQ_LDS_ATOMIC_RETURN", 0) / idx, 1), int(avg.get("SQ_INSTS_LDS", 0)), int(avg["GRBM_GUI_ACTIVE"] / 8)])
print(open(sys.argv[2]).read())
