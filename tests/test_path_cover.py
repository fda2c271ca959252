"""The path cover behind the plan-guided walk (kbo_amd/csrc/path_cover.cpp), checked on the CPU against the index's
own subset matrix: every row sits at exactly one position, and text[p] != 0 certifies a real edge of the de Bruijn
graph from node_at[p-1] to node_at[p] with that label (reference semantics of extend-right: SURVEY.md section 8(a) A0)."""
import numpy as np
import pytest

import kbo_amd
from kbo_amd import synth


def _check_cover(sbwt):
    n, k = sbwt.n_sets(), sbwt.k()
    rows, Carr, lcs = sbwt.export_parts()
    text, pos, node = sbwt.path_cover()
    assert sorted(pos.tolist()) == list(range(n)) and np.array_equal(node[pos], np.arange(n, dtype=np.uint32))
    bits = [np.unpackbits(r.view(np.uint8), bitorder="little")[:n].astype(np.int64) for r in rows]
    rank = [np.concatenate([[0], np.cumsum(b)]) for b in bits]  # rank[c][i] = set bits of B_c in [0, i)
    first = np.ones(n, dtype=bool)
    first[1:] = lcs[1:].astype(np.int64) < k - 1  # rows that open their (k-1)-suffix group
    group = np.maximum.accumulate(np.where(first, np.arange(n), 0))
    edges = 0
    for p in range(1, n):
        ch = int(text[p])
        if ch == 0:
            continue
        c = b"ACGT".index(ch)
        g = int(group[node[p - 1]])
        assert bits[c][g] == 1, (p, "no such edge")
        assert Carr[c] + rank[c][g] == node[p], (p, "edge leads elsewhere")
        edges += 1
    assert text[0] == 0
    return edges


@pytest.mark.parametrize("k", [1, 2, 3, 5, 31, 40])
def test_path_cover_invariants(k):
    rng = np.random.default_rng(k)
    g = synth.genome(3000, seed=90 + k)
    rep = np.tile(g[:40], 6)
    seqs = [np.concatenate([g, rep]).tobytes(), g[100:900].tobytes() + b"N" + g[5:400].tobytes(), b"ACGACGACGACGACGACG" * 3,
            bytes(rng.choice(list(b"AC"), 300).astype(np.uint8))]
    sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, num_threads=2))
    edges = _check_cover(sbwt)
    if k >= 5:
        assert edges >= sbwt.n_sets() // 2  # the cover is made of long paths, not of single nodes


def test_path_cover_single_contig_is_one_path():
    g = synth.genome(20000, seed=5)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=2))
    text, pos, node = sbwt.path_cover()
    assert int((text == 0).sum()) <= 3  # root + dummy chain + genome: one path, give or take a repeated k-mer
    assert _check_cover(sbwt) >= sbwt.n_sets() - 3


@pytest.mark.parametrize("k,add_revcomp", [(5, False), (31, False), (31, True), (200, False)])
def test_recovery_lines_hold_rank_blocks_and_lcs(k, add_revcomp):
    """kbo_index_recovery_lines: every 128-byte line = four rank blocks {C[c] + rank_c(64 b), 64 row bits} + 64 LCS bytes,
    checked against the index's own parts; rows beyond the last one carry no bits and LCS 0; the last line is all zero."""
    rng = np.random.default_rng(11 + k)
    seqs = [bytes(rng.choice(list(b"ACGT"), 5000).astype(np.uint8)), b"ACGT" * 300 + b"N" + bytes(rng.choice(list(b"ACGT"), 700).astype(np.uint8))]
    sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, add_revcomp=add_revcomp, num_threads=2))
    n = sbwt.n_sets()
    rows, Carr, lcs = sbwt.export_parts()
    lines = sbwt.recovery_lines()
    assert lines.shape == (n // 64 + 3, 128)
    assert not lines[-1].any()
    bits = [np.unpackbits(np.asarray(r, dtype=np.uint64).view(np.uint8), bitorder="little")[:n] for r in rows]
    cum = [np.concatenate([[0], np.cumsum(b)]) for b in bits]
    for b in range(n // 64 + 2):
        lo, hi = 64 * b, min(n, 64 * b + 64)
        for c in range(4):
            blk = lines[b, 16 * c:16 * c + 16].view(np.uint32)
            assert blk[0] == int(Carr[c]) + int(cum[c][min(lo, n)])
            want = np.zeros(64, dtype=np.uint8)
            if lo < n:
                want[:hi - lo] = bits[c][lo:hi]
            assert np.array_equal(np.unpackbits(blk[1:3].view(np.uint8), bitorder="little"), want)
            assert blk[3] == 0
        want_lcs = np.zeros(64, dtype=np.uint8)
        if lo < n:
            want_lcs[:hi - lo] = lcs[lo:hi]
        assert np.array_equal(lines[b, 64:], want_lcs)


def test_contraction_from_the_lines_windows_equals_the_entries():
    """The recovery kernel takes a contraction level out of two 16-row LCS windows of the lines ([.., l] and [r, ..], cut at
    the line's ends) instead of the {lcs, psv, nsv} entries.  Model of both in numpy on a real index: wherever the windows
    hold the level's ends the two agree, and they fall short exactly when an end lies outside them."""
    rng = np.random.default_rng(5)
    g = synth.genome(30_000, seed=9)
    rep = np.tile(g[:300], 6)
    sbwt, _ = kbo_amd.build([np.concatenate([g, rep]).tobytes()], kbo_amd.BuildOpts(k=31, num_threads=2))
    n = sbwt.n_sets()
    _, _, lcs = sbwt.export_parts()
    lines = sbwt.recovery_lines()
    lcs_at = lambda i: int(lcs[i]) if i < n else 0  # noqa: E731  (sentinel behind the last row)

    def entries(l, r):  # device_index.cpp / sbwt_index.hpp: one level up with psv / nsv
        lv = max(lcs_at(l), lcs_at(r))
        if lv == 0:
            return 0, 0, n
        nl, nr = l, r
        if lcs_at(l) == lv:
            nl = l - 1
            while lcs_at(nl) >= lv:
                nl -= 1
        if lcs_at(r) == lv:
            nr = r + 1
            while lcs_at(nr) >= lv:
                nr += 1
        return lv, nl, nr

    def windows(l, r):  # plan_kernels.hip, ms_walk_recovery_kernel
        bl, br, ol, orr = l >> 6, r >> 6, l & 63, r & 63
        wl, wr = (ol - 15 if ol > 15 else 0), min(orr, 48)
        wa = lines[bl, 64 + wl:64 + wl + 16].astype(int)
        wb = lines[br, 64 + wr:64 + wr + 16].astype(int)
        pl, pr = ol - wl, orr - wr
        lv = max(wa[pl], wb[pr])
        if lv == 0:
            return 0, 0, n
        nl, nr = l, r
        if wa[pl] == lv:
            below = [q for q in range(pl) if wa[q] < lv]
            if not below:
                return None
            nl = (bl << 6) + wl + below[-1]
        if wb[pr] == lv:
            above = [q for q in range(pr + 1, 16) if wb[q] < lv]
            if not above:
                return None
            nr = (br << 6) + wr + above[0]
        return lv, nl, nr

    short = 0
    for _ in range(20_000):
        l = int(rng.integers(0, n))
        r = min(n, l + int(rng.choice([1, 1, 1, 2, 3, 8, 40])))
        want = entries(l, r)
        got = windows(l, r)
        if got is None:
            short += 1
            lv, nl, nr = want
            bl, br = l >> 6, r >> 6
            wl_abs = (bl << 6) + ((l & 63) - 15 if (l & 63) > 15 else 0)
            wr_abs = (br << 6) + min(r & 63, 48) + 15
            assert nl < wl_abs or nr > wr_abs
        else:
            assert got == want
    assert 0 < short < 6_000
