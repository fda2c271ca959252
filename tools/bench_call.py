#!/usr/bin/env python3
"""kbo call over a batch of long reads, the WHOLE call (first pass on the device, site windows, second pass on host
threads, variants packed): kbo_call_batch through the C ABI, timed per read; a sample of reads against the oracle's literal
kbo::call.  Usage: tools/bench_call.py [G=5000000] [R=50000] [L=10000] [K=51] (environment)."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd  # noqa: E402
from kbo_amd import batch, synth  # noqa: E402

G, R, L, K = (int(os.environ.get(n, d)) for n, d in (("G", 5_000_000), ("R", 50_000), ("L", 10_000), ("K", 51)))
CHECK = int(os.environ.get("CHECK", 200))
g = synth.genome(G)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=K, num_threads=16))
sbwt.to_device()
if os.environ.get("SLAB_MB"):
    kbo_amd.lib().kbo_set_slab_bytes(int(os.environ["SLAB_MB"]) << 20)
rng = np.random.default_rng(7)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
t0 = time.perf_counter()
starts = rng.integers(0, G - L - 64, R)
reads = np.empty((R, L), dtype=np.uint8)
for r in range(R):
    reads[r] = g[starts[r]:starts[r] + L]
hit = rng.random((R, L)) < 0.01                      # substitutions
reads[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
for r in range(0, R, 3):                              # a deletion and an insertion in every third read
    p = int(rng.integers(200, L - 200))
    reads[r, p:L - 3] = reads[r, p + 3:].copy()
    q = int(rng.integers(200, L - 200))
    reads[r, q + 2:] = reads[r, q:L - 2].copy()
    reads[r, q:q + 2] = acgt[rng.integers(0, 4, 2)]
concat = reads.reshape(-1)
offsets = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
print(f"reads made in {time.perf_counter() - t0:.1f} s", flush=True)
opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=K, build_select=True))
best, res = 1e9, None
for it in range(3):
    t0 = time.perf_counter()
    res = batch.call_batch_arrays(sbwt, concat, offsets, opts)
    dt = time.perf_counter() - t0
    best = min(best, dt)
    print(f"kbo_call_batch: {dt * 1e3:.1f} ms for {R} x {L} bp reads = {dt / R * 1e6:.2f} us/read, {R * L / dt / 1e9:.2f} Gbp/s, "
          f"{int(res['var_offsets'][-1])} variants", flush=True)
print(f"best: {best / R * 1e6:.2f} us/read ({0.47e-3 / (best / R):.1f} x round 2's 0.47 ms/read)")
if CHECK:
    from oracle import binding as ora
    rows, Carr, lcs = sbwt.export_parts()
    oi = ora.Index.from_parts(K, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
    bad = 0
    for s in rng.integers(0, R, CHECK):
        exp, _, _ = oi.call(reads[s].tobytes(), K, 1e-7)
        got = [(p, q.decode(), r.decode()) for p, q, r in batch.variants_of(res, int(s))]
        bad += got != exp
    print(f"{CHECK} reads against oracle.call: {'all equal' if not bad else str(bad) + ' DIFFER'}")
    sys.exit(1 if bad else 0)
