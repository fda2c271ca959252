"""Shared pieces of the -m gpu tests (imported by test modules; holds no test itself)."""
import os

import numpy as np

import kbo_amd


def threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def adopt(oracle, sbwt):
    """The oracle over the product-built index (its own row-sorting builder needs minutes beyond ~20 Mbp; builder
    equality is tests/test_builder_vs_oracle.py)."""
    rows, Carr, lcs = sbwt.export_parts()
    return oracle.Index.from_parts(sbwt.k(), sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)


def call_walk_sites(L, sbwt, dev, thr):
    """kbo_call_walk_dev over a DeviceBatch -> (set of (offset of i, offset of j, row), MS bytes, usable)"""
    import torch
    lists = 256  # KBO_CALL_LISTS
    cap = (dev.total // 4 + 8192) // lists * lists
    sites = torch.zeros((cap, 4), dtype=torch.int32, device=dev.device)
    count = torch.zeros(lists * 16 + 16, dtype=torch.int32, device=dev.device)
    s = torch.cuda.current_stream(dev.device)
    dev.ms.fill_(0xEE)
    kbo_amd.check(L.kbo_call_walk_dev(sbwt._h, dev.q.data_ptr(), dev.off.data_ptr(), dev.n_seqs, dev.total, dev.max_len, thr,
                                      dev.ms.data_ptr(), sites.data_ptr(), cap, count.data_ptr(), dev.work.data_ptr(),
                                      dev.work_bytes, s.cuda_stream))
    torch.cuda.synchronize()
    c = count.cpu().numpy()
    seg = cap // lists
    ok = bool((c[:lists * 16:16] <= seg).all()) and int(c[lists * 16]) == 0
    h = sites.cpu().numpy().view(np.uint32)
    raw = np.concatenate([h[g * seg:g * seg + min(int(c[g * 16]), seg)] for g in range(lists)])
    raw = raw[raw[:, 0] != 0xFFFFFFFF]
    return {(int(a), int(b), int(r)) for a, b, r, _ in raw}, dev.ms[:dev.total].cpu().numpy(), ok


def oracle_sites(ora, concat, offsets, thr):
    """variant_calling.rs:266-273 for every read (oracle, literal) as the set call_walk_sites returns"""
    recs = ora.call_sites_batch(concat, offsets, thr, n_threads=threads())
    off = np.asarray(offsets, dtype=np.uint64)
    base = off[recs[:, 0].astype(np.int64)]
    return {(int(b + i), int(b + j), int(r)) for b, i, j, r in zip(base, recs[:, 1], recs[:, 2], recs[:, 3])}


def long_reads(rng, g, n_reads, read_len, sub_rate):
    """reads of read_len bases off g with substitutions (numpy; synth.reads is for 150 bp reads at scale)"""
    starts = rng.integers(0, len(g) - read_len, n_reads)
    out = np.empty(n_reads * read_len, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for r, a in enumerate(starts):
        p = g[a:a + read_len].copy()
        hit = rng.random(read_len) < sub_rate
        p[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
        out[r * read_len:(r + 1) * read_len] = p
    return out, np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
