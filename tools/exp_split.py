#!/usr/bin/env python3
"""Would the A1 stage gain from running plan_kernel of one half of a batch under the guided walk of the other half?
Two half-batches on two streams, the second delayed by about one plan kernel, against the whole batch on one stream."""
import os, sys, time
import numpy as np, torch
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd
from kbo_amd import batch, synth
G = int(os.environ.get("G", 5_000_000)); R = int(os.environ.get("R", 1_000_000)); K = int(os.environ.get("K", 40))
PARTS = int(os.environ.get("PARTS", 2)); DELAY = int(os.environ.get("DELAY", 200_000))
g = synth.genome(G)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
concat, offsets = synth.reads(g, R, 150, 0.01)
dev = torch.device("cuda:0")
whole = batch.DeviceBatch(sbwt, concat, offsets, device=dev)
per = R // PARTS
parts = [batch.DeviceBatch(sbwt, concat[p * per * 150:(p + 1) * per * 150], offsets[:per + 1], device=dev) for p in range(PARTS)]
streams = [torch.cuda.Stream(dev) for _ in range(PARTS)]
whole.walk(streams[0]); [p.walk(streams[0]) for p in parts]
torch.cuda.synchronize()
ref = whole.ms[:whole.total].clone()

def serial():
    for _ in range(K):
        whole.walk(streams[0])

def split(delay):
    for i, st in enumerate(streams):
        if i and delay:
            with torch.cuda.stream(st):
                torch.cuda._sleep(delay * i)
    for _ in range(K):
        for p, st in zip(parts, streams):
            p.walk(st)

for name, fn in (("whole batch, one stream", serial), (f"{PARTS} parts, {PARTS} streams, in phase", lambda: split(0)),
                 (f"{PARTS} parts, {PARTS} streams, staggered", lambda: split(DELAY)), ("whole batch, one stream", serial),
                 (f"{PARTS} parts, {PARTS} streams, staggered", lambda: split(DELAY))):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    got = torch.cat([p.ms[:p.total] for p in parts])
    print(f"{name}: {dt / K * 1e3:.3f} ms per A1 stage over {R} reads, equal: {bool(torch.equal(got, ref[:got.numel()]))}", flush=True)
