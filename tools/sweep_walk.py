#!/usr/bin/env python3
"""GPU-side tuning sweep of the walk kernel launch geometry on the C2 workload.
Prints walk-kernel ms (min / median over repeats) per configuration."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd  # noqa: E402
from kbo_amd import batch, synth  # noqa: E402

G = int(os.environ.get("G", 5_000_000))
R = int(os.environ.get("R", 1_000_000))
g = synth.genome(G)
if os.environ.get("PAIRS"):
    # two-base steps on/off (and the depth from which they are tried): each setting needs a fresh device copy
    concat, offsets = synth.reads(g, R, 150, float(os.environ.get("SUB", 0.01)))
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=32))
    from kbo_amd import index as kindex
    path = "/tmp/sweep_index"
    kindex.save_flat(path, sbwt)
    base = None
    for setting in os.environ["PAIRS"].split(","):
        min_depth = int(setting)
        kbo_amd.lib().kbo_set_pair_steps((1 << 63) if min_depth < 0 else 0, max(min_depth, 0))
        sbwt, _ = kindex.load_flat(path)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
        stream = torch.cuda.current_stream()
        ts = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); dev.walk(stream); b.record(stream)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = dev.ms.clone()
        if base is None:
            base = ms
        print(f"G={G} pairs {'off' if min_depth < 0 else 'from depth %d' % min_depth}: walk min {min(ts[1:]):.3f} ms median {float(np.median(ts[1:])):.3f}"
              f"  -> {R*150/min(ts[1:])/1e6:.1f} Gbp/s  same MS as first: {bool(torch.equal(ms, base))}", flush=True)
        del dev, sbwt
    sys.exit(0)
import time as _t
_t0 = _t.time()
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=32))
print("build s", _t.time() - _t0, "n_sets", sbwt.n_sets(), "device bytes", sbwt.device_bytes(), flush=True)
concat, offsets = synth.reads(g, R, 150, float(os.environ.get("SUB", 0.01)))
dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
stream = torch.cuda.current_stream()


def time_walk(reps=8):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); dev.walk(stream); b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), float(np.median(ts))


L = kbo_amd.lib()
if os.environ.get("NOPLAN"):  # the plain walk (default: whatever kbo_ms_batch_dev takes, i.e. the plan-guided walk)
    L.kbo_set_plan(0, 0, 0)
if os.environ.get("RARE"):
    for wpc in (32,):
        for period in (2, 4, 6, 8, 12, 16, 32):
            L.kbo_set_walk_threads(64); L.kbo_set_walk_waves_per_cu(wpc); L.kbo_set_walk_rare(period)
            dev.walk(stream); torch.cuda.synchronize()
            mn, med = time_walk()
            print(f"waves/CU={wpc} period={period:2d}  walk min {mn:.3f} ms median {med:.3f}", flush=True)
    sys.exit(0)
if os.environ.get("ONLY"):
    L.kbo_set_walk_threads(64); L.kbo_set_walk_waves_per_cu(32)
    dev.walk(stream); torch.cuda.synchronize()
    print("only:", time_walk(4))
    sys.exit(0)
for threads in (64, 256):
    for wpc in (8, 12, 16, 20, 24, 28, 32):
        L.kbo_set_walk_threads(threads)
        L.kbo_set_walk_waves_per_cu(wpc)
        dev.walk(stream); torch.cuda.synchronize()
        mn, med = time_walk()
        print(f"threads={threads:3d} waves/CU={wpc:2d}  walk min {mn:.3f} ms  median {med:.3f} ms  -> {R*150/mn/1e6:.1f} Gbp/s", flush=True)
