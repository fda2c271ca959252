"""The committed evidence hangs together (pure JSON, no GPU): the traffic figure bench.py quotes is the fold of the committed
rocprofv3 summary by the guide's rule - (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch, TCC_MISS_sum misses -, carries a
build fingerprint, and the committed C2 bench line quotes exactly that figure for exactly that build."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEY = "5000000x1000000x150x0.01:map"


def _load(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.load(f)


def _round_tag():
    """'r04' from the entry's source, 'profiles/r04_c2_summary.json (...)': the files of the round that folded it"""
    src = _load("traffic_latest.json")["workloads"][KEY]["source"]
    name = os.path.basename(src.split()[0])
    assert name.endswith("_c2_summary.json"), src
    return name[:-len("_c2_summary.json")]


def test_traffic_is_the_fold_of_the_committed_summary():
    entry = _load("traffic_latest.json")["workloads"][KEY]
    pmc = _load(_round_tag() + "_c2_summary.json")["pmc_avg_per_launch"]["map_reads_kernel"]
    assert entry["a1_bytes_per_launch"] == int((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
    assert entry["a1_tcc_miss_per_launch"] == int(pmc["TCC_MISS_sum"])
    assert len(entry["build_sha16"]) == 16


def test_committed_bench_line_quotes_that_traffic_for_that_build():
    entry = _load("traffic_latest.json")["workloads"][KEY]
    line = _load(_round_tag() + "_bench_c2.json")
    ro = line["roofline"]
    assert ro["build_sha16"] == entry["build_sha16"] and ro["traffic"] == entry["a1_bytes_per_launch"]
    assert ro["l2_miss_per_launch"] == entry["a1_tcc_miss_per_launch"]
    # the line's own arithmetic: bytes per launch over the device's time per launch, and the figures printed beside it
    bytes_per_launch = ro["algorithmic_bytes_per_base"] * ro["units_per_launch"]
    assert abs(bytes_per_launch / (ro["device_ms_per_launch"] * 1e-3) / 1e9 - ro["achieved"]) / ro["achieved"] < 2e-3
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert ro["launches_sharing_the_device"] == 2 and ro["device_ms_per_launch"] >= ro["kernel_ms"] / 2
    assert abs(ro["device_ms_per_launch"] - line["ms_per_step"]) < 1e-3
    assert ro["per_launch_duration"]["frac"] < ro["frac"] and 0 < ro["alone"]["frac"] < 1
    assert abs(ro["wasted_traffic"] - ro["traffic"] / bytes_per_launch) < 5e-3
    assert line["bit_exact_vs_oracle"] is True and all(v["bit_exact_vs_oracle"] for v in line["sensitivity"])
    # kernel_stats of the same command: the kernel's average over its dispatches lies between the kernel alone and the live figure
    stats = open(os.path.join(ROOT, "profiles", _round_tag() + "_c2_kernel_stats.csv")).read().splitlines()
    row = next(r for r in stats if "map_reads_kernel" in r)
    avg_ms = float(row.rsplit('"', 1)[1].split(",")[3]) / 1e6
    assert ro["alone"]["kernel_ms"] * 0.9 < avg_ms < ro["kernel_ms"] * 1.1
