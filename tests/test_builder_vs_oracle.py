"""Product SBWT builder (kbo_amd/csrc/sbwt_build.cpp, via the C ABI) against the oracle's
independent row-sorting builder: identical subset-matrix rows, C array, LCS array, n_sets
and n_kmers.  CPU only (no compute kernels are called)."""
import os

import numpy as np
import pytest

import kbo_amd
from kbo_amd import synth


def _same_index(prod, ora):
    rows, C, lcs = prod.export_parts()
    assert (prod.k(), prod.n_sets(), prod.n_kmers()) == (ora.k, ora.n_sets, ora.n_kmers)
    assert C == ora.C
    for c in range(4):
        assert np.array_equal(rows[c], ora.bits(c)), f"row {c}"
    assert np.array_equal(lcs, ora.lcs())


def _rand_seqs(rng, n, lo, hi, with_n=True):
    seqs = []
    for _ in range(n):
        s = rng.choice(list(b"ACGT"), size=int(rng.integers(lo, hi))).astype(np.uint8)
        if with_n and len(s) > 10 and rng.random() < 0.5:
            s[int(rng.integers(0, len(s)))] = ord("N")
        seqs.append(s.tobytes())
    return seqs


@pytest.mark.parametrize("k", [1, 2, 3, 5, 9, 16, 31, 32, 33, 51, 63, 64, 65, 100, 129])
def test_builder_matches_oracle_random(oracle, k):
    rng = np.random.default_rng(1000 + k)
    seqs = _rand_seqs(rng, 4, max(k, 5), 4 * k + 200)
    seqs.append(seqs[0][: len(seqs[0]) // 2])  # duplicated k-mers
    seqs.append(b"ACG")                        # shorter than k for most k
    prod, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k))
    _same_index(prod, oracle.Index.build(seqs, k=k))


@pytest.mark.parametrize("k", [3, 7, 31, 40])
def test_builder_revcomp(oracle, k):
    rng = np.random.default_rng(77 + k)
    seqs = _rand_seqs(rng, 3, k + 5, 300)
    prod, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, add_revcomp=True))
    _same_index(prod, oracle.Index.build(seqs, k=k, add_revcomp=True))


def test_builder_low_complexity_and_threads(oracle):
    seqs = [b"A" * 200, b"ACACACACACACACACACACACACAC" * 8, b"T" * 50 + b"G" * 50]
    for nt in (1, 4):
        prod, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=11, num_threads=nt))
        _same_index(prod, oracle.Index.build(seqs, k=11))


def test_builder_threaded_large(oracle):
    g = synth.genome(300_000)
    prod, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=4))
    _same_index(prod, oracle.Index.build([g.tobytes()], k=31))
    assert prod.n_sets() == prod.n_kmers() + 31  # single random contig: k dummy rows


def test_reference_example_index(golden):
    g = golden["ms"][0]
    prod, _ = kbo_amd.build(g["ref_seqs"], kbo_amd.BuildOpts(k=g["k"]))
    assert (prod.n_sets(), prod.n_kmers()) == (g["n_sets"], g["n_kmers"])
    f = golden["find"][0]
    prod, _ = kbo_amd.build(f["ref_seqs"], kbo_amd.BuildOpts(k=f["k"]))
    assert (prod.n_sets(), prod.n_kmers()) == (f["n_sets"], f["n_kmers"])


def test_save_load_from_parts_roundtrip(tmp_path, oracle):
    """index.rs:277-296 tests serialisation as a round trip only; same here."""
    seqs = [b"AAAGAACCA-TCAGGGCG", b"GATTACAGATTACATTTGGGA"]
    prod, lcs = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=4))
    prefix = os.path.join(tmp_path, "serialized_index_test")
    kbo_amd.index.serialize_sbwt(prefix, prod, lcs)
    loaded, _ = kbo_amd.index.load_sbwt(prefix)
    ora = oracle.Index.build(seqs, k=4)
    _same_index(loaded, ora)
    rows, C, l = prod.export_parts()
    again = kbo_amd.SbwtIndexVariant.from_parts(4, prod.n_sets(), prod.n_kmers(), rows, C, l)
    _same_index(again, ora)
    with pytest.raises(kbo_amd.KboError) as e:
        kbo_amd.index.load_sbwt(os.path.join(tmp_path, "does_not_exist"))
    assert e.value.code == -10


def test_device_layout_sizes():
    prod, _ = kbo_amd.build([synth.genome(50_000)], kbo_amd.BuildOpts(k=31))
    rank_bytes, lcs_bytes = prod.device_bytes()
    n = prod.n_sets()
    assert rank_bytes == (n // 96 + 2) * 16 * 4
    assert lcs_bytes == (3 * (n + 1) + 4) * 4
