#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory into a small JSON summary:
per-kernel stats (calls, avg/min/max ns) and per-kernel PMC values averaged per launch."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    m = re.search(r"(map_reads_kernel|\w+_kernel|__amd_rocclr_\w+|vectorized_elementwise_kernel)", name)
    return m.group(1) if m else name[:40]


summary = {"kernel_stats": [], "pmc_avg_per_launch": {}}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        summary["kernel_stats"].append({"kernel": short(row["Name"]), "calls": int(row["Calls"]),
                                        "avg_ns": float(row["AverageNs"]), "min_ns": float(row["MinNs"]),
                                        "max_ns": float(row["MaxNs"]), "pct": float(row["Percentage"])})
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(set))
    meta = {}
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k][row["Counter_Name"]].add(row["Dispatch_Id"])
        meta[k] = {"vgpr": row["VGPR_Count"], "sgpr": row["SGPR_Count"], "lds": row["LDS_Block_Size"],
                   "grid": row["Grid_Size"], "wg": row["Workgroup_Size"]}
    for k in acc:
        if "elementwise" in k or "rocclr" in k:
            continue
        d = summary["pmc_avg_per_launch"].setdefault(k, {"launch_meta": meta[k]})
        for c in acc[k]:
            d[c] = acc[k][c] / max(1, len(disp[k][c]))
for k, d in summary["pmc_avg_per_launch"].items():
    if "FETCH_SIZE" in d:
        d["FETCH_bytes(KBx1024; gfx950 may under-read 2x)"] = d["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in d:
        d["WRITE_bytes"] = d["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in d:
        d["L2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
print(json.dumps(summary, indent=1))
