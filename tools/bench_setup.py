#!/usr/bin/env python3
"""What the plan structures of a device copy cost and when they pay (device_index.cpp plan_break_even_bases):
for a few index sizes the set-up seconds of the copy by part (kbo_index_device_layout), the time of one resident batch of 150-base
reads through the plain walk and through kbo_map_batch_dev's one kernel, and the bases after which the structures have paid for
themselves - with the path cover laid out by this copy, and with a cover that came with the handle (an index file).
python tools/bench_setup.py [genome sizes, comma separated]"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kbo_amd  # noqa: E402
from kbo_amd import batch, synth  # noqa: E402

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1000000,5000000,25000000,100000000").split(",")]
L = kbo_amd.lib()
device = torch.device("cuda:0")
stream = torch.cuda.current_stream(device)
print("| index | rows | set-up: layout + upload / cover / lines / seed / tables (s) | plain walk Gbp/s | one kernel Gbp/s | pays after (own cover) | pays after (stored cover) |")
print("|---|---|---|---|---|---|---|")
for G in sizes:
    g = synth.genome(G)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
    R = 1_000_000
    concat, offsets = synth.reads(g, R, 150, 0.01)

    def rate(dev, steps=10):
        for _ in range(2):
            dev.run(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            dev.run(stream)
        torch.cuda.synchronize()
        return dev.total * steps / (time.perf_counter() - t0)

    L.kbo_set_plan(0, 0, 0)  # a copy without plan structures: the plain walk
    L.kbo_set_plan_lazy(1 << 62)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=device, format=True, want_ms=False)
    plain = rate(dev)
    del dev
    L.kbo_set_plan(1, 0, 0)
    L.kbo_set_plan_lazy(-1)
    sb2, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
    t0 = time.perf_counter()
    dev = batch.DeviceBatch(sb2, concat, offsets, device=device, format=True, want_ms=False)  # (to_device: everything at once)
    t_copy = time.perf_counter() - t0
    one = rate(dev)
    lay = sb2.device_layout()
    base_s = lay["layout_seconds"] + lay["upload_seconds"]
    plan_s = lay["cover_seconds"] + lay["lines_seconds"] + lay["seed_seconds"] + lay["dtab_seconds"]
    gain = 1.0 / plain - 1.0 / one  # seconds saved per base
    pays = plan_s / gain if gain > 0 else float("inf")
    pays_stored = (plan_s - lay["cover_seconds"]) / gain if gain > 0 else float("inf")
    print(f"| {G / 1e6:g} Mbp | {sb2.n_sets()} | {base_s:.2f} / {lay['cover_seconds']:.2f} / {lay['lines_seconds']:.2f} / {lay['seed_seconds']:.2f} / {lay['dtab_seconds']:.2f} "
          f"| {plain / 1e9:.1f} | {one / 1e9:.1f} ({'one kernel' if dev.fused else 'two kernels'}) | {pays / 1e9:.1f} Gbp | {pays_stored / 1e9:.1f} Gbp |", flush=True)
    del dev, sb2, sbwt
