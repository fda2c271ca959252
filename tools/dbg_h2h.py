"""Where bench.py's host-to-host leg loses its rate: the same leg after each stage of what bench.py does before it."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, ROOT)
import bench
import kbo_amd
from kbo_amd import batch, synth
args = bench.parse(["--no-extras"])
g = synth.genome(args.genome)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
sbwt.to_device()


def leg(tag):
    r = bench.host_to_host_leg(args, sbwt, g)
    print(tag, r["value"], r["packed"]["value"], flush=True)


leg("fresh index")
import torch
dev = torch.device("cuda:0")
concat, offsets = synth.reads(g, 1_000_000, 150, 0.01)
d = batch.DeviceBatch(sbwt, concat, offsets, device=dev, format=True, want_ms=False)
for _ in range(20):
    d.run()
torch.cuda.synchronize()
leg("after torch + DeviceBatch runs")
from oracle import binding as ora
rows, Carr, lcs = sbwt.export_parts()
oi = ora.Index.from_parts(31, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
leg("after the oracle index exists")
exp = oi.matches_batch(concat, offsets, 1e-7, n_threads=16)
leg("after oracle threads ran")
t = oi.matches_batch_timed(concat, offsets, 1e-7, n_threads=16, passes=2) if hasattr(oi, "matches_batch_timed") else None
leg("after the pinned oracle pool ran")
L = kbo_amd.lib()
L.kbo_set_plan_stats(1)
d.run(); torch.cuda.synchronize(); d.walk(); torch.cuda.synchronize()
L.kbo_set_plan_stats(0)
leg("after a counting launch (kbo_set_plan_stats)")
L.kbo_set_stage_timing(1)
d.run(); torch.cuda.synchronize()
L.kbo_set_stage_timing(0)
import ctypes as C
a, b, n = C.c_double(), C.c_double(), C.c_int()
L.kbo_stage_timing_read(C.byref(a), C.byref(b), C.byref(n))
leg("after stage timing")
tail = torch.cuda.Stream(dev)
sens = bench.sensitivity_leg(args, g, sbwt, oi, torch, dev, torch.cuda.current_stream(dev), tail)
leg("after the sensitivity leg")
