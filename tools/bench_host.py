#!/usr/bin/env python3
"""End-to-end host->host rate of kbo_map_batch on the C2 workload (reads and output in host
memory; PCIe-inclusive).  Not the bench.py metric — reported in DESIGN.md §7."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd  # noqa: E402
from kbo_amd import batch, synth  # noqa: E402

G, R = int(os.environ.get("G", 5_000_000)), int(os.environ.get("R", 4_000_000))
g = synth.genome(G)
sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=16))
sbwt.to_device()
concat, offsets = synth.reads(g, R, 150, 0.01)
out = np.zeros(len(concat), dtype=np.uint8)  # allocated and touched once, outside the timed region
L = kbo_amd.lib()
PINNED = bool(os.environ.get("PINNED"))  # the caller's buffers in pinned memory (hipHostMalloc): used in place, no staging copies


def pinned_like(a):
    import torch
    t = torch.empty(a.shape, dtype=getattr(torch, str(a.dtype)), pin_memory=True)
    v = t.numpy()
    v[...] = a
    _KEEP.append(t)
    return v


_KEEP = []
if PINNED:
    concat, out = pinned_like(concat), pinned_like(out)
if os.environ.get("PACKED"):  # 2-bit words in, 2-bit words / run lengths out (kbo_matches_batch_packed, kbo_find_batch_packed)
    import ctypes as C
    from kbo_amd import _capi
    from oracle import binding as ora
    words, pos, byt = batch.pack_reads(concat, offsets)
    wout = np.zeros(len(words), dtype=np.uint32)
    if PINNED:
        words, wout = pinned_like(words), pinned_like(wout)
    for slab_mb in [int(x) for x in os.environ.get('SLABS', '32,64,128').split(',')]:
        L.kbo_set_slab_bytes(slab_mb << 20)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            kbo_amd.check(L.kbo_matches_batch_packed(sbwt._h, words.ctypes.data, offsets.ctypes.data, R, None, None, 0, 1e-7, wout.ctypes.data))
            best = min(best, time.perf_counter() - t0)
        print(f"slab {slab_mb:4d} MiB: matches, packed {R * 150 / best / 1e9:6.2f} Gbp/s host->host ({best * 1e3:.1f} ms for {R * 150 / 1e6:.0f} Mbp, "
              f"{len(words) * 4 / 1e6:.0f} MB each way)", flush=True)
        co = _capi.FindOpts(1e-7, 0)
        ro = np.zeros(R + 1, dtype=np.uint64)
        best = 1e9
        for _ in range(5):
            p = C.c_void_p()
            t0 = time.perf_counter()
            kbo_amd.check(L.kbo_find_batch_packed(sbwt._h, words.ctypes.data, offsets.ctypes.data, R, None, None, 0, C.byref(co), C.byref(p), ro.ctypes.data))
            best = min(best, time.perf_counter() - t0)
            L.kbo_free(p)
        print(f"slab {slab_mb:4d} MiB: find,    packed {R * 150 / best / 1e9:6.2f} Gbp/s host->host ({best * 1e3:.1f} ms, {int(ro[-1])} runs)", flush=True)
    rows, Carr, lcs = sbwt.export_parts()
    oi = ora.Index.from_parts(31, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
    n_chk = min(R, 200_000)
    exp = oi.matches_batch(concat[:n_chk * 150], offsets[:n_chk + 1], 1e-7, n_threads=16)
    got = batch.unpack_matches(wout, offsets)
    print("packed output equals the oracle on the first", n_chk, "reads:", bool(np.array_equal(got[:n_chk * 150], exp)))
    sys.exit(0)
for slab_mb in [int(x) for x in os.environ.get('SLABS', '32,64,128,256').split(',')]:
    L.kbo_set_slab_bytes(slab_mb << 20)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        kbo_amd.check(L.kbo_map_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, R, 1e-7, 1, out.ctypes.data))
        best = min(best, time.perf_counter() - t0)
    print(f"slab {slab_mb:4d} MiB: {R * 150 / best / 1e9:6.2f} Gbp/s host->host ({best * 1e3:.1f} ms for {R * 150 / 1e6:.0f} Mbp)", flush=True)
    if os.environ.get("MS"):  # A1 only: MS values back, then MS values + intervals
        d = np.zeros(len(concat), dtype=np.uint8)
        lo = np.zeros(len(concat), dtype=np.uint32)
        hi = np.zeros(len(concat), dtype=np.uint32)
        for what, a, b in (("ms", None, None), ("ms + intervals", lo.ctypes.data, hi.ctypes.data)):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                kbo_amd.check(L.kbo_ms_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, R, d.ctypes.data, a, b))
                best = min(best, time.perf_counter() - t0)
            print(f"{what:>15s}: {R * 150 / best / 1e9:6.2f} Gbp/s host->host ({best * 1e3:.1f} ms)", flush=True)
    if os.environ.get("FIND"):  # kbo::find: run lengths come back instead of characters
        import ctypes as C
        from kbo_amd import _capi
        co = _capi.FindOpts(1e-7, 0)
        ro = np.zeros(R + 1, dtype=np.uint64)
        best = 1e9
        for _ in range(4):
            p = C.POINTER(_capi.RLE)()
            t0 = time.perf_counter()
            kbo_amd.check(L.kbo_find_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, R, C.byref(co), C.byref(p), ro.ctypes.data))
            best = min(best, time.perf_counter() - t0)
            L.kbo_free(p)
        print(f"           find: {R * 150 / best / 1e9:6.2f} Gbp/s host->host ({best * 1e3:.1f} ms, {int(ro[-1])} runs for {R} reads)", flush=True)
        buf = np.zeros((int(ro[-1]) + 1024, 7), dtype=np.uint64)  # caller-owned records, reused from call to call
        n_runs = C.c_size_t()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            kbo_amd.check(L.kbo_find_batch_into(sbwt._h, concat.ctypes.data, offsets.ctypes.data, R, C.byref(co), buf.ctypes.data,
                                                len(buf), ro.ctypes.data, C.byref(n_runs)))
            best = min(best, time.perf_counter() - t0)
        print(f"      find_into: {R * 150 / best / 1e9:6.2f} Gbp/s host->host ({best * 1e3:.1f} ms)", flush=True)
