"""kbo_map_batch_dev on the C2 index with the insertion / deletion reads of bench.py's sensitivity leg: counters of the kernel."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kbo_amd, bench
from kbo_amd import batch, synth
g = synth.genome(5_000_000)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
L = kbo_amd.lib()
dev0 = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev0)
for name, (concat, offsets) in (("subs", synth.reads(g, 1_000_000, 150, 0.01)), ("indel", bench.indel_reads(g, 1_000_000, 150, 0.01, 0.002, seed=0x5E11C))):
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=True, want_ms=False)
    L.kbo_set_plan_stats(1)
    dev.run(); torch.cuda.synchronize()
    st = dev.plan_stats()
    L.kbo_set_plan_stats(0)
    for _ in range(3): dev.run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(10): dev.run()
    b.record(stream); torch.cuda.synchronize()
    print(name, "ms/step %.4f" % (a.elapsed_time(b) / 10), {k: st[k] for k in ("seed_lookups", "mismatches", "tab_lookups", "tab_flagged", "tab_anchored", "items_noplan")}, flush=True)
