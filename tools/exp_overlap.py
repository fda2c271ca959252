#!/usr/bin/env python3
"""A5/A6 of batch i on a second stream while A1 of batch i+1 runs (C2 workload): step time with and without."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd
from kbo_amd import batch, synth
G = int(os.environ.get("G", 5_000_000)); R = int(os.environ.get("R", 1_000_000)); K = int(os.environ.get("K", 40))
g = synth.genome(G)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
concat, offsets = synth.reads(g, R, 150, 0.01)
dev = torch.device("cuda:0")
B = [batch.DeviceBatch(sbwt, concat, offsets, device=dev, format=True) for _ in range(2)]
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for b in B:
    b.run(sA)
torch.cuda.synchronize()
ref = B[0].chars.clone()

def serial():
    for i in range(K):
        B[i % 2].run(sA)

def overlapped():
    ev1 = [torch.cuda.Event() for _ in range(2)]; ev2 = [torch.cuda.Event() for _ in range(2)]
    for i in range(K):
        b = B[i % 2]
        if i >= 2: sA.wait_event(ev2[i % 2])
        b.walk(sA); ev1[i % 2].record(sA)
        sB.wait_event(ev1[i % 2]); b.derand_translate(sB); ev2[i % 2].record(sB)

for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = bool(torch.equal(B[0].chars, ref) and torch.equal(B[1].chars, ref))
    print(f"{name}: {dt / K * 1e3:.3f} ms per step, {R * 150 / (dt / K) / 1e9:.1f} Gbp/s, chars equal: {ok}", flush=True)
