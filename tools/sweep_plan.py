#!/usr/bin/env python3
"""Plan-guided walk vs plain walk on synthetic reads: walk time (plan kernel + guided kernel) and MS equality.
G=genome bases, R=reads, SUB=substitution rates (comma separated), DMIN / CAP lists to sweep, GW = list of
<guided waves per CU>:<recovery lines 0/1>."""
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
L = kbo_amd.lib()
g = synth.genome(G)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=32))
print("n_sets", sbwt.n_sets(), flush=True)
stream = torch.cuda.current_stream()


def time_walk(dev, reps=8):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); dev.walk(stream); b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), float(np.median(ts))


for sub in [float(x) for x in os.environ.get("SUB", "0.01").split(",")]:
    concat, offsets = synth.reads(g, R, 150, sub)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
    L.kbo_set_plan(0, 0, 0)
    dev.ms.zero_(); dev.walk(stream); torch.cuda.synchronize()
    base = dev.ms.clone()
    mn, med = time_walk(dev)
    print(f"sub={sub}: plain walk min {mn:.3f} ms median {med:.3f}  ({R*150/mn/1e6:.1f} Gbp/s)", flush=True)
    for dmin in [int(x) for x in os.environ.get("DMIN", "-1").split(",")]:
        for cap in [int(x) for x in os.environ.get("CAP", "64").split(",")]:
         for gw, fatl in [(int(x.split(":")[0]), int(x.split(":")[1])) for x in os.environ.get("GW", "8:0").split(",")]:
          L.kbo_set_guided_walk(gw, fatl)
          for wpc in [int(x) for x in os.environ.get("WPC", "32").split(",")]:
            for rare in [int(x) for x in os.environ.get("RARE", "8").split(",")]:
                L.kbo_set_plan(1, dmin, cap); L.kbo_set_walk_rare(rare); L.kbo_set_walk_waves_per_cu(wpc)
                dev.ms.fill_(0xEE); dev.walk(stream); torch.cuda.synchronize()
                same = bool(torch.equal(dev.ms[:dev.total], base[:dev.total]))
                if not same:
                    bad = torch.nonzero(dev.ms[:dev.total] != base[:dev.total]).flatten()
                    print("   MISMATCH at", bad[:10].tolist(), "of", int(bad.numel()),
                          "got", dev.ms[bad[:10]].tolist(), "want", base[bad[:10]].tolist(), flush=True)
                mn, med = time_walk(dev)
                print(f"   plan dmin={dmin} cap={cap} rare={rare} wpc={wpc} guided waves/CU={gw} recovery lines={fatl}: walk min {mn:.3f} ms median {med:.3f}  "
                      f"({R*150/mn/1e6:.1f} Gbp/s)  same MS: {same}", flush=True)
    L.kbo_set_walk_rare(8); L.kbo_set_walk_waves_per_cu(32)
    del dev
