"""Pins the CPU oracle (oracle/kbo_oracle.c) against every golden vector the
reference's own tests hold for the hot path.  CPU only."""
import math

import numpy as np
import pytest


def test_ms_golden(golden, oracle):  # index.rs:265-273
    for g in golden["ms"]:
        idx = oracle.Index.build(g["ref_seqs"], k=g["k"])
        assert idx.n_sets == g["n_sets"] and idx.n_kmers == g["n_kmers"]
        d, lo, hi = idx.matching_statistics(g["query"])
        assert d.tolist() == g["expected_ms"]
        assert np.all(lo < hi)


def test_appendix_a_index(oracle):
    """rows, bits, LCS, C and intervals of the index.rs:265 example."""
    idx = oracle.Index.build(["AAAGAACCA-TCAGGGCG"], k=3)
    rows = [idx.access_kmer(i).decode() for i in range(idx.n_sets)]
    assert rows == ["$$$", "AAA", "GAA", "CCA", "TCA", "AGA", "AAC", "ACC", "GGC", "$TC", "AAG",
                    "CAG", "GCG", "AGG", "GGG", "$$T"]
    assert idx.C == [1, 6, 10, 15]
    assert idx.lcs().tolist() == [0, 0, 2, 1, 2, 1, 0, 1, 1, 1, 0, 2, 1, 1, 2, 0]
    bits = [[(int(idx.bits(c)[0]) >> i) & 1 for i in range(16)] for c in range(4)]
    assert bits[0] == [0, 1, 0, 0, 0, 1, 0, 1, 0, 1, 1, 0, 0, 0, 0, 0]
    assert bits[1] == [0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 1]
    assert bits[2] == [0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 1, 0, 0]
    assert bits[3] == [1] + [0] * 15
    assert sum(map(sum, bits)) == idx.n_sets - 1
    d, lo, hi = idx.matching_statistics("CAAGCCACTCATTGGGTC")
    exp = [(1, 6, 10), (2, 3, 5), (2, 1, 3), (3, 10, 11), (2, 8, 9), (2, 7, 8), (3, 3, 4), (2, 6, 7),
           (1, 15, 16), (2, 9, 10), (3, 4, 5), (1, 15, 16), (1, 15, 16), (1, 10, 15), (2, 13, 15),
           (3, 14, 15), (1, 15, 16), (2, 9, 10)]
    assert list(zip(d.tolist(), lo.tolist(), hi.tolist())) == exp


def test_log_rm_max_cdf(golden, oracle):  # derandomize.rs:298-304, :77-79
    g = golden["log_rm_max_cdf"]
    for t, e in zip(g["t"], g["expected"]):
        assert abs(oracle.log_rm_max_cdf(t, g["alphabet_size"], g["n_kmers"]) - e) < g["tol"]
    dt = g["doctest"]
    assert abs(oracle.log_rm_max_cdf(dt["t"], dt["alphabet_size"], dt["n_kmers"]) - dt["expected"]) < dt["tol"]


def test_random_match_threshold(golden, oracle):  # derandomize.rs:307-314
    g = golden["random_match_threshold"]
    got = [oracle.random_match_threshold(g["k"], g["n_kmers"], g["alphabet_size"], p)
           for p in g["max_error_prob"]]
    assert got == g["expected"]
    # thresholds quoted in BASELINE.md §3 for the bench configs (k=31, p=1e-7)
    assert [oracle.random_match_threshold(31, n, 4, 1e-7) for n in
            (10**6, 5 * 10**6, 10**8, 25 * 10**7, 3 * 10**9)] == [21, 22, 24, 25, 27]


def test_derandomize(golden, oracle):  # derandomize.rs:317-379
    for g in golden["derandomize_ms_val"]:
        assert oracle.derandomize_ms_val(*g["args"]) == g["expected"]
    for g in golden["derandomize_ms_vec"]:
        assert oracle.derandomize_ms_vec(g["noisy_ms"], g["k"], g["threshold"]).tolist() == g["expected"]


def test_translate(golden, oracle):  # translate.rs:396-532
    for g in golden["translate_ms_val"]:
        assert list(oracle.translate_ms_val(*g["args"])) == g["expected"]
    for g in golden["translate_ms_vec"]:
        assert oracle.translate_ms_vec(g["derand_ms"], g["k"], g["threshold"]) == g["expected"]


def test_asserts_mirror_reference(oracle):
    """len<=2 / threshold<=1 / empty query panic in the reference
    (derandomize.rs:274-276, translate.rs:268-270, index.rs:248)."""
    with pytest.raises(oracle.OracleError) as e:
        oracle.derandomize_ms_vec([1, 2], 3, 2)
    assert e.value.code == -2
    with pytest.raises(oracle.OracleError) as e:
        oracle.derandomize_ms_vec([1, 2, 3], 3, 1)
    assert e.value.code == -3
    with pytest.raises(oracle.OracleError) as e:
        oracle.translate_ms_vec([1, 2], 3, 2)
    assert e.value.code == -2
    idx = oracle.Index.build(["ACGTACGT"], k=3)
    with pytest.raises(oracle.OracleError) as e:
        idx.matching_statistics(b"")
    assert e.value.code == -1


def test_matches_golden(golden, oracle):  # lib.rs:600-609
    for g in golden["matches"]:
        idx = oracle.Index.build(g["ref_seqs"], k=g["k"])
        assert idx.matches(g["query"], g["max_error_prob"]).decode() == g["expected"]


def test_map_golden_no_refinement(golden, oracle):  # lib.rs:670-717
    for g in golden["map"]:
        if g["fill_gaps"] or g["call_variants"]:
            continue
        idx = oracle.Index.build(g["query_seqs"], k=g["k"])
        aln = idx.matches(g["ref_seq"], g["max_error_prob"])
        out = oracle.relative_to_ref(g["ref_seq"], aln) if g["format"] else aln
        assert out.decode() == g["expected"]


def test_find_golden(golden, oracle):  # lib.rs:786-805 (k=31, threshold from n_kmers, gapped RLE)
    for g in golden["find"]:
        idx = oracle.Index.build(g["ref_seqs"], k=g["k"])
        assert (idx.n_kmers, idx.n_sets) == (g["n_kmers"], g["n_sets"])
        assert oracle.random_match_threshold(g["k"], idx.n_kmers, 4, g["max_error_prob"]) == g["threshold"]
        aln = idx.matches(g["query"], g["max_error_prob"])
        got = oracle.run_lengths_gapped(aln, g["max_gap_len"])
        assert [list(r) for r in got] == g["expected"]


def test_run_lengths_golden(golden, oracle):  # format.rs:295-330
    for g in golden["run_lengths"]:
        got = oracle.run_lengths_gapped(g["aln"], g["max_gap_len"])
        assert [list(r) for r in got] == g["expected"]


def test_nearest_unique_context_golden(golden, oracle):
    """gap_filling.rs:535-564: pins colex intervals + access_kmer (row content)."""
    for g in golden["nearest_unique_context"]:
        idx = oracle.Index.build([g["query"]], k=g["k"])
        d, lo, hi = idx.matching_statistics(g["reference"])
        a, b = g["search_range"]
        i = b
        while i >= a and hi[i] - lo[i] != 1:
            i -= 1
        assert [i, idx.access_kmer(int(lo[i])).decode()] == g["expected"]


# ---------------------------------------------------------------- definitional checks

def _brute_ms(rows_set_suffixes, q, k):
    out = []
    for i in range(len(q)):
        best = 0
        for L in range(1, min(k, i + 1) + 1):
            if q[i + 1 - L:i + 1] in rows_set_suffixes:
                best = L
            else:
                break
        out.append(best)
    return out


@pytest.mark.parametrize("seed", range(6))
def test_ms_matches_definition(oracle, seed):
    """d_i = longest suffix of q[..=i], <= k, that is a suffix of some SBWT row
    ($ never matches), and [lo,hi) = exactly the rows having that suffix."""
    rng = np.random.default_rng(seed)
    k = int(rng.integers(2, 9))
    refs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(k, 60)))) for _ in range(3)]
    refs[1] = refs[1][:len(refs[1]) // 2] + "N" + refs[1][len(refs[1]) // 2:]
    idx = oracle.Index.build(refs, k=k)
    rows = [idx.access_kmer(i).decode() for i in range(idx.n_sets)]
    assert rows == sorted(set(rows), key=lambda r: ["$ACGT".index(c) for c in reversed(r)])
    suffixes = set()
    for r in rows:
        s = r.lstrip("$")
        for L in range(1, len(s) + 1):
            suffixes.add(s[len(s) - L:])
    q = "".join(rng.choice(list("ACGT"), size=80))
    q = q[:40] + refs[0][:20] + q[40:]
    d, lo, hi = idx.matching_statistics(q)
    assert d.tolist() == _brute_ms(suffixes, q, k)
    for i in range(len(q)):
        di = int(d[i])
        want = [j for j, r in enumerate(rows) if di == 0 or r.lstrip("$").endswith(q[i + 1 - di:i + 1])] \
            if di > 0 else list(range(idx.n_sets))
        assert want == list(range(int(lo[i]), int(hi[i])))


def test_k_minus_1_warmup_property(oracle):
    """a walk restarted from the empty state k-1 bases upstream gives
    identical (d, lo, hi)."""
    rng = np.random.default_rng(7)
    for k in (3, 5, 11, 31):
        ref = "".join(rng.choice(list("ACGT"), size=3000))
        idx = oracle.Index.build([ref], k=k)
        q = list(ref[500:1100])
        for p in rng.integers(0, len(q), size=30):
            q[p] = "ACGT"[(("ACGT".index(q[p])) + 1) % 4]
        q = "".join(q)
        d, lo, hi = idx.matching_statistics(q)
        for s in rng.integers(k, len(q) - 1, size=25):
            w = int(s) - (k - 1)
            d2, lo2, hi2 = idx.matching_statistics(q[w:])
            off = int(s) - w
            assert d2[off:].tolist() == d[s:].tolist()
            assert lo2[off:].tolist() == lo[s:].tolist() and hi2[off:].tolist() == hi[s:].tolist()


def test_revcomp_index(oracle):
    """add_revcomp (index.rs:76,89; untested upstream): index of seq with revcomp ==
    index of [seq, revcomp(seq)]."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    s = "ACGGTCAGGTTTACCAGT"
    rc = "".join(comp[c] for c in reversed(s))
    a = oracle.Index.build([s], k=5, add_revcomp=True)
    b = oracle.Index.build([s, rc], k=5)
    assert a.n_sets == b.n_sets and a.n_kmers == b.n_kmers
    assert all(np.array_equal(a.bits(c), b.bits(c)) for c in range(4))
    assert np.array_equal(a.lcs(), b.lcs())


# ---------------------------------------------------------------- refinement stages (oracle/kbo_oracle_refine.c)

def test_oracle_call_goldens(golden, oracle):  # variant_calling.rs:312-454, lib.rs:526-544
    for g in golden["call_variants"]:
        idx = oracle.Index.build([g["reference"]], k=g["k"])
        got, _, _ = idx.call(g["query"], g["k"], g["max_error_prob"])
        assert [list(v) for v in got] == g["expected"], g["src"]
    for g in golden["call"]:
        idx = oracle.Index.build([g["query"]], k=g["k"])
        got, _, _ = idx.call(g["reference"], g["k"], g["max_error_prob"])
        assert [list(v) for v in got] == g["expected"], g["src"]


def test_oracle_add_variants_goldens(golden, oracle):  # translate.rs:324-347, 535-676
    for g in golden["add_variants"]:
        k, t = g["k"], g["threshold"]
        idx = oracle.Index.build([g["query"]], k=k)
        d, _, _ = idx.matching_statistics(g["reference"])
        tr = oracle.translate_ms_vec(oracle.derandomize_ms_vec(d, k, t), k, t)
        _, buf, n = idx.call(g["reference"], k, g["max_error_prob"])
        assert oracle.add_variants(tr, buf, n).decode() == g["expected"], g["src"]


def test_oracle_fill_gaps_goldens(golden, oracle):  # gap_filling.rs:419-441, 641-922
    for g in golden["fill_gaps"]:
        idx = oracle.Index.build([g["query"]], k=g["k"])
        t = g["threshold"] if g["threshold"] is not None else \
            oracle.random_match_threshold(g["k"], idx.n_kmers, 4, g["max_err_prob"])
        d, _, _ = idx.matching_statistics(g["reference"])
        tr = oracle.translate_ms_vec(oracle.derandomize_ms_vec(d, g["k"], t), g["k"], t)
        assert idx.fill_gaps(tr, g["reference"], t, g["max_err_prob"]).decode() == g["expected"], g["src"]


def test_oracle_map_goldens(golden, oracle):  # lib.rs:647-717
    for g in golden["map"]:
        idx = oracle.Index.build(g["query_seqs"], k=g["k"])
        got = idx.map(g["ref_seq"], g["k"], g["max_error_prob"], g["fill_gaps"], g["call_variants"], g["format"])
        assert got.decode() == g["expected"], g["src"]


def test_access_kmer_of_an_adopted_index(oracle):
    """An index adopted from its parts (ora_index_from_parts) has no row table: access_kmer spells the row from the subset
    matrix alone (the extend-right bijection read backwards).  Same k-mers as the built index, hence the same kbo::call."""
    rng = np.random.default_rng(5)
    for k in (3, 5, 31, 64):
        seqs = [bytes(rng.choice(list(b"ACGT"), int(rng.integers(300, 4000))).astype(np.uint8)) for _ in range(3)]
        seqs[1] = seqs[1][:40] + b"N" + seqs[1][40:] + seqs[0][:100]
        a = oracle.Index.build(seqs, k=k)
        b = oracle.Index.from_parts(k, a.n_sets, a.n_kmers, [a.bits(c) for c in range(4)], a.C, a.lcs())
        for i in range(a.n_sets):
            assert a.access_kmer(i) == b.access_kmer(i), (k, i)
        rd = bytearray(seqs[2][:1500])
        for p in (300, 700, 1200):
            if p < len(rd):
                rd[p] = ord("A") if rd[p] != ord("A") else ord("C")
        assert a.call(bytes(rd), k, 1e-3)[0] == b.call(bytes(rd), k, 1e-3)[0]
    a = oracle.Index.build([b"AAAGAACCA-TCAGGGCG"], k=3)  # the reference's index (index.rs:265): rows as in SURVEY appendix A
    b = oracle.Index.from_parts(3, a.n_sets, a.n_kmers, [a.bits(c) for c in range(4)], a.C, a.lcs())
    assert [b.access_kmer(i) for i in range(16)] == [a.access_kmer(i) for i in range(16)]
    assert b.access_kmer(9) == b"$TC" and b.access_kmer(15) == b"$$T"
