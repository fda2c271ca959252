"""The figures of bench.py lines one reads first: value, ms per step, roofline.frac, parity; kernel alone / second pass; the variants; the
MS-emitting entry points.  python tools/show_line.py <line.json> [..]"""
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["bit_exact_vs_oracle"])
        print("  one at a time", d["one_batch_at_a_time"]["value"], d["one_batch_at_a_time"]["map_reads_kernel_ms"], d["one_batch_at_a_time"]["second_pass_ms"])
        if d.get("sensitivity"): print("  ", [(v["variant"][:12], round(v["value"]/1000), v["bit_exact_vs_oracle"]) for v in d["sensitivity"]])
        if d.get("ms_variant"): m=d["ms_variant"]; print("  ms", m["kbo_ms_batch_dev"]["value"], m["kbo_ms_batch_dev"]["two_streams"]["value"], m["kbo_map_batch_dev_want_ms"]["value"])
    except Exception as e: print(f, "ERR", e)
