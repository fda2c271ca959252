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
