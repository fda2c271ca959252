#!/usr/bin/env python3
"""Two experiments on the plain walk kernel at the C2 shape:
 (a) lanes per wave that take reads (64 / 32 / 16 / 8): what a tiling that gives several lanes to one read would have to beat;
 (b) LDS reserved per wave (0 / 5 / 10 / 20 KB, untouched): what staging a wave's MS bytes for a fused A5/A6 costs."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd  # noqa: E402
from kbo_amd import batch, synth  # noqa: E402

L = kbo_amd.lib()
L.kbo_set_plan(0, 0, 0)
g = synth.genome(5_000_000)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
stream = torch.cuda.current_stream()


def timed(dev, reps=6):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); dev.walk(stream); b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:])


for lanes in (64, 32, 16, 8):
    R = 1_000_000 * lanes // 64  # the same number of waves and of reads per working lane
    concat, offsets = synth.reads(g, R, 150, 0.01)
    # the item list is laid out for 64 lanes per wave: give every wave 64 slots of which `lanes` hold reads
    pad = np.zeros((R // lanes, 64 - lanes, 150), dtype=np.uint8) + ord("A")
    full = np.concatenate([concat.reshape(R // lanes, lanes, 150), pad], axis=1).reshape(-1)
    off = np.arange(R // lanes * 64 + 1, dtype=np.uint64) * 150
    dev = batch.DeviceBatch(sbwt, full, off, device=torch.device("cuda:0"))
    L.kbo_set_walk_experiment(lanes, 0)
    t = timed(dev)
    print(f"lanes with reads {lanes:2d}: walk {t:.3f} ms for {R} reads = {t / R * 1e6:.2f} ms per million reads", flush=True)
    del dev
L.kbo_set_walk_experiment(64, 0)
concat, offsets = synth.reads(g, 1_000_000, 150, 0.01)
dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
for lds in (0, 5 << 10, 10 << 10, 20 << 10, 40 << 10):
    L.kbo_set_walk_experiment(64, lds)
    print(f"LDS reserved per wave {lds >> 10:2d} KB: walk {timed(dev):.3f} ms", flush=True)
L.kbo_set_walk_experiment(64, 0)
