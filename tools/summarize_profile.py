#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel stats + PMC averages for the
scoring kernel + HBM traffic per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md section HBM
prescribes for gfx950 wide streaming reads; WRITE_SIZE as read; both are in KiB)."""
import collections
import csv
import json
import os
import sys

d = sys.argv[1]
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
with open(os.path.join(d, "trace", "run_kernel_stats.csv")) as f:
    for row in csv.DictReader(f):
        print(f'{row["Name"][:70]:70s} calls {row["Calls"]:>4s} avg_ns {float(row["AverageNs"]):12.1f} '
              f'pct {row["Percentage"]}')
vals = {}
kname = None
for p in ("pmc_a", "pmc_b", "pmc_c", "pmc_d"):
    path = os.path.join(d, p, "run_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "score_multi" in r["Kernel_Name"] or "score_uniform" in r["Kernel_Name"]:
            kname = r["Kernel_Name"].split("(")[0]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            vals["VGPR_Count"] = r.get("VGPR_Count")
            vals["SGPR_Count"] = r.get("SGPR_Count")
            vals["LDS_Block_Size"] = r.get("LDS_Block_Size")
    for c, v in acc.items():
        vals[c] = sum(v) / len(v)
print("== scoring kernel PMC averages per launch ==")
for k in sorted(vals):
    print(f"{k:24s} {vals[k]}")
if "FETCH_SIZE" in vals:
    fetch = vals["FETCH_SIZE"] * 1024 * 2  # gfx950: FETCH_SIZE reports half of a wide stream
    write = vals.get("WRITE_SIZE", 0.0) * 1024
    out = {"kernel": kname, "fetch_bytes_corrected": fetch, "write_bytes": write,
           "hbm_bytes_per_launch": fetch + write, "source": "profile (rocprofv3 --pmc, tools/profile.sh)",
           "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB counters, "
                   "separate --pmc passes"}
    print("== HBM traffic per launch ==")
    print(json.dumps(out))
    json.dump(out, open(os.path.join(d, "traffic.json"), "w"))
