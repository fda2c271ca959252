#!/usr/bin/env python3
"""Folds a tools/summarize_prof.py summary into profiles/traffic_latest.json (what bench.py quotes as roofline.traffic):
A1-stage bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 and TCC misses, summed over the stage's kernels, per launch.
Usage: tools/traffic_from_summary.py <summary.json> <workload key, e.g. 5000000x1000000x150x0.01:plan> <source label> [launches per step]
(launches per step: resident slabs per GPU of the workload - C3's 10 M reads are two slabs - so that the figures are per STEP of
bench.py, like its stage time)"""
import json
import os
import sys

MAP = ("map_reads_kernel",)  # workload keys that end in ":map": the one kernel of kbo_map_batch_dev, priced on its own
A1 = ("plan_kernel", "dtab_resolve_kernel", "dtab_stretch_kernel", "plan_count_kernel", "scan_kernel", "plan_emit_kernel", "ms_walk_guided_kernel",
      "ms_walk_recovery_kernel", "redo_collect_kernel", "ms_walk_kernel", "call_fix_sites_kernel", "make_chunk_items_kernel", "chunk_count_kernel")
summ, key, source = json.load(open(sys.argv[1])), sys.argv[2], sys.argv[3]
per_step_launches = int(sys.argv[4]) if len(sys.argv) > 4 else 1
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "traffic_latest.json")
doc = json.load(open(path))
calls = {k["kernel"]: k["calls"] for k in summ["kernel_stats"]}
is_map = key.endswith(":map")
per_step = min(c for k, c in calls.items() if k in (("map_reads_kernel",) if is_map else ("plan_kernel", "ms_walk_kernel")))  # launches of the stage in the run
kern, tot_b, tot_m = {}, 0, 0
for k in (MAP if is_map else A1):
    d = summ["pmc_avg_per_launch"].get(k)
    if not d or "FETCH_SIZE" not in d:
        continue
    mult = calls.get(k, per_step) / per_step if k == "scan_kernel" else 1  # (several scan launches per stage)
    mult *= per_step_launches
    b = int((2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0)) * 1024 * mult)
    m = int(d.get("TCC_MISS_sum", 0) * mult)
    kern[k] = {"bytes": b, "tcc_miss": m, "tcc_hit": int(d.get("TCC_HIT_sum", 0) * mult)}
    tot_b += b
    tot_m += m
sys.path.insert(0, root)
import bench  # noqa: E402  (build_sha16: what this profile was taken of)
doc["workloads"][key] = {"source": source, "launches_per_step": per_step_launches, "a1_bytes_per_launch": tot_b,
                         "a1_tcc_miss_per_launch": tot_m, "kernels": kern, "build_sha16": bench.build_sha16()}  # (per step of bench.py = per launch x launches per step)
json.dump(doc, open(path, "w"), indent=1)
print(key, "bytes", tot_b, "misses", tot_m)
