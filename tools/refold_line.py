#!/usr/bin/env python3
"""Fills the traffic fields of a committed bench line (--config C5: benchlib/call.py's roofline) from profiles/traffic_latest.json, as
bench.py itself does when the fold is there before the run - for a line whose profiler passes ran AFTER it in the same session
(tools/profile_c5.sh: the index is built once, the line first).  Usage: tools/refold_line.py <line.json> <workload key>"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line_path, key = sys.argv[1], sys.argv[2]
line = json.load(open(line_path))
entry = json.load(open(os.path.join(root, "profiles", "traffic_latest.json")))["workloads"][key]
ro = line["roofline"]
own = sum(ro["bytes_by_part"].values())
ro["traffic"] = entry["a1_bytes_per_launch"]
ro["traffic_source"] = entry["source"] + " (passes of the same session as this line: tools/profile_c5.sh, folded in by tools/refold_line.py)"
ro["wasted_traffic"] = round(ro["traffic"] / own, 2)
ro["traffic_frac"] = round(ro["traffic"] / (ro["kernel_ms"] * 1e-3) / 8e12, 4)
ro["l2_miss_per_launch"] = entry["a1_tcc_miss_per_launch"]
ro["traffic_by_kernel"] = entry["kernels"]
json.dump(line, open(line_path, "w"))
print(key, "traffic", ro["traffic"], "wasted", ro["wasted_traffic"], "traffic_frac", ro["traffic_frac"])
