"""Shared pieces of the -m gpu tests (imported by test modules; holds no test itself)."""
import os

import numpy as np

import kbo_amd


def threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def adopt(oracle, sbwt, parts=False):
    """The oracle over the product-built index (its own row-sorting builder needs minutes beyond ~20 Mbp; builder
    equality is tests/test_builder_vs_oracle.py up to 300 kbp, check_rows_off_the_text at any size)."""
    rows, Carr, lcs = sbwt.export_parts()
    ora = oracle.Index.from_parts(sbwt.k(), sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
    return (ora, lcs) if parts else ora


def check_rows_off_the_text(oracle, ora, lcs, sbwt, g, n_samples=10_000, seed=7):
    """A check of the index that passes through NEITHER builder (the oracle adopts the product-built index beyond 20 Mbp):
    `g` is one ACGT contig whose k-mers are all distinct (n_sets == len(g) + 1 says so: G - k + 1 k-mers + k dummy rows).
    For a sample of k-mers taken off the text, oracle/index_check.c counts - by a pass over the text alone - how many of
    the text's k-mers are colex-smaller; with the dummy rows `$..$ g[0:j]` that are smaller (direct string comparison) that
    IS the k-mer's row in the index index.rs:56-99 defines.  Checked against it: (1) the interval the product's walk of
    the k-mer ends in (depth k, that single row); (2) the k-mer the subset matrix spells at that row and at its two
    neighbours (oracle access_kmer over the adopted parts: k backward steps over the matrix, no row table): the row's own
    k-mer is the sampled one, the neighbours' ranks counted off the text are row - 1 and row + 1 and they occur in the
    text exactly once; (3) the LCS values of those rows against the common suffixes of the spelled k-mers."""
    from kbo_amd import batch
    k, n, G = sbwt.k(), sbwt.n_sets(), len(g)
    assert n == G + 1
    rng = np.random.default_rng(seed)
    pos = np.unique(np.concatenate([[0, 1, G - k], rng.integers(0, G - k + 1, n_samples)]))
    kmers = np.stack([g[p:p + k] for p in pos])
    less, eq = oracle.kmer_colex_ranks(g, k, kmers, n_threads=threads())
    assert (eq == 1).all()
    head = g[:k].tobytes()
    rev_heads = [head[:j][::-1] for j in range(k)]  # dummy row j = $^(k-j) g[0:j]: smaller iff rev(g[0:j]) <= the k-mer's last j characters reversed

    def dummies_below(km):
        r = km[::-1]
        return sum(1 for j in range(k) if rev_heads[j] <= r[:j])
    rows = np.array([int(l) + dummies_below(km.tobytes()) for l, km in zip(less, kmers)], dtype=np.int64)
    # (1) the product's walk with intervals over the sampled k-mers, one sequence each
    off = np.arange(len(kmers) + 1, dtype=np.uint64) * np.uint64(k)
    d, lo, hi = batch.ms_batch(sbwt, kmers.ravel(), off, want_intervals=True)
    last = (off[1:] - np.uint64(1)).astype(np.int64)
    assert (d[last] == k).all()
    assert np.array_equal(lo[last].astype(np.int64), rows) and np.array_equal(hi[last].astype(np.int64), rows + 1)
    # (2), (3) on a part of the sample (k select steps per spelled row)
    sub = rng.choice(len(rows), size=min(len(rows), 1500), replace=False)
    spelled, want_rank = [], []
    for i in sub:
        r = int(rows[i])
        assert ora.access_kmer(r) == kmers[i].tobytes()
        for rr in (r - 1, r + 1):
            if 0 <= rr < n:
                s = ora.access_kmer(rr)
                if b"$" not in s:
                    spelled.append(np.frombuffer(s, dtype=np.uint8))
                    want_rank.append(rr)
        for rr in (r, r + 1):  # LCS[rr] = longest common suffix of rows rr - 1 and rr ($ matches nothing)
            if 1 <= rr < n:
                a, b = ora.access_kmer(rr - 1), ora.access_kmer(rr)
                c = 0
                while c < k and a[k - 1 - c] == b[k - 1 - c] and a[k - 1 - c] != ord("$"):
                    c += 1
                assert int(lcs[rr]) == c, (rr, a, b)
    if spelled:
        sp = np.stack(spelled)
        l2, e2 = oracle.kmer_colex_ranks(g, k, sp, n_threads=threads())
        assert (e2 == 1).all()
        assert [int(l) + dummies_below(km.tobytes()) for l, km in zip(l2, sp)] == want_rank


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
