"""C-ABI checks that need no GPU: the library loads, exports every symbol that
include/kbo_hip.h declares, the host-only entry points (A3 threshold maths, scalar
*_val functions, format) match the reference's goldens, argument checks mirror the
reference's asserts, and compute entry points fail loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import kbo_amd
from kbo_amd import _capi, derandomize, format, translate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    L = C.CDLL(_capi.LIB_PATH)
    for header, symbols in (("kbo_hip.h", _capi.SYMBOLS), ("kbo_hip_tuning.h", _capi.TUNING_SYMBOLS)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        declared = set(re.findall(r"\b(kbo_[a-z0-9_]+)\s*\(", hdr))
        assert declared == set(symbols), (header, declared ^ set(symbols))
        for name in declared:
            assert hasattr(L, name), name
    # the drop-in boundary carries no tuning knob or test hook
    assert not (set(_capi.SYMBOLS) & set(_capi.TUNING_SYMBOLS))
    assert not [n for n in _capi.SYMBOLS if "experiment" in n or "force" in n or "tuning" in n]


def test_no_oracle_in_product():
    """The product must not link or import the oracle."""
    import subprocess
    out = subprocess.run(["ldd", _capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kbo_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "kbo_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_opts_defaults_match_reference():
    # lib.rs:300-313, 344-353, 373-382, 398-407, 454-466
    b = _capi.BuildOpts(); kbo_amd.lib().kbo_build_opts_default(C.byref(b))
    assert (b.k, b.add_revcomp, b.num_threads, b.prefix_precalc, b.build_select, b.mem_gb,
            b.dedup_batches, b.temp_dir) == (31, 0, 1, 8, 0, 4, 0, None)
    f = _capi.FindOpts(); kbo_amd.lib().kbo_find_opts_default(C.byref(f))
    assert (f.max_error_prob, f.max_gap_len) == (0.0000001, 0)
    m = _capi.MapOpts(); kbo_amd.lib().kbo_map_opts_default(C.byref(m))
    assert (m.max_error_prob, m.fill_gaps, m.call_variants, m.format, m.sbwt_build_opts.build_select,
            m.sbwt_build_opts.k) == (0.0000001, 1, 1, 1, 1, 31)
    assert kbo_amd.BuildOpts() == kbo_amd.BuildOpts(31, False, 1, 8, False, 4, False, None)
    assert kbo_amd.CallOpts().sbwt_build_opts.build_select and kbo_amd.MapOpts().fill_gaps
    assert kbo_amd.MatchOpts().max_error_prob == kbo_amd.FindOpts().max_error_prob == 1e-7


def test_log_rm_max_cdf_golden(golden):  # derandomize.rs:298-304
    g = golden["log_rm_max_cdf"]
    for t, e in zip(g["t"], g["expected"]):
        assert abs(derandomize.log_rm_max_cdf(t, g["alphabet_size"], g["n_kmers"]) - e) < g["tol"]


def test_random_match_threshold_golden(golden, oracle):  # derandomize.rs:307-314
    g = golden["random_match_threshold"]
    assert [derandomize.random_match_threshold(g["k"], g["n_kmers"], g["alphabet_size"], p)
            for p in g["max_error_prob"]] == g["expected"]
    rng = np.random.default_rng(3)
    for _ in range(300):  # product (C++) and oracle (C) restatements agree everywhere
        k = int(rng.integers(2, 256)); n = int(rng.integers(1, 10**10)); p = float(10 ** -rng.uniform(0, 12))
        assert derandomize.random_match_threshold(k, n, 4, p) == oracle.random_match_threshold(k, n, 4, p)
    for bad in [(0, 10, 4, 0.1), (31, 0, 4, 0.1), (31, 10, 0, 0.1), (31, 10, 4, 0.0), (31, 10, 4, 1.5)]:
        with pytest.raises(kbo_amd.KboError) as e:  # derandomize.rs:133-137
            derandomize.random_match_threshold(*bad)
        assert e.value.code == -4


def test_scalar_vals_golden(golden):
    for g in golden["derandomize_ms_val"]:  # derandomize.rs:317-370
        assert derandomize.derandomize_ms_val(*g["args"]) == g["expected"]
    for g in golden["translate_ms_val"]:  # translate.rs:396-498
        assert list(translate.translate_ms_val(*g["args"])) == g["expected"]
    with pytest.raises(kbo_amd.KboError) as e:  # derandomize.rs:229
        derandomize.derandomize_ms_val(5, 1, 2, 3)
    assert e.value.code == -9
    with pytest.raises(kbo_amd.KboError) as e:  # translate.rs:186
        translate.translate_ms_val(1, 1, 1, 1)
    assert e.value.code == -3


def test_format_golden(golden, oracle):  # format.rs:295-330
    for g in golden["run_lengths"]:
        got = format.run_lengths(g["aln"])
        assert [[r.start, r.end, r.matches, r.mismatches, r.jumps, r.gap_bases, r.gap_opens] for r in got] \
            == g["expected"]
    rng = np.random.default_rng(5)
    for _ in range(200):  # differential: product host RLE vs oracle restatement
        aln = "".join(rng.choice(list("MMMMMM--XRR-"), size=int(rng.integers(1, 120))))
        if aln[0] == "R":
            aln = "M" + aln[1:]
        for gap in (0, 1, 3, 50):
            got = [(r.start, r.end, r.matches, r.mismatches, r.jumps, r.gap_bases, r.gap_opens)
                   for r in format.run_lengths_gapped(aln, gap)]
            assert got == oracle.run_lengths_gapped(aln, gap)
        ref = "".join(rng.choice(list("ACGT"), size=len(aln)))
        assert format.relative_to_ref(ref, aln) == oracle.relative_to_ref(ref, aln)
    # an alignment that starts with 'R': format.rs:175 evaluates aln[i - 1] with i = 0 and panics; mirrored as KBO_E_REF_PANIC
    for aln in ("RRM", "R", "RM-M"):
        with pytest.raises(kbo_amd.KboError) as e:
            format.run_lengths_gapped(aln, 0)
        assert e.value.code == -11
    assert [r.start for r in format.run_lengths_gapped("-RRM", 0)] == [1]  # (behind a gap it is an ordinary run)
    assert format.relative_to_ref("ACGTAC", "MRIXD-") == b"ACG---"
    assert format.relative_to_ref("ACGT", "MGNM") == b"AGNT"


def test_vector_arg_checks_need_no_gpu():
    """len<=2 / threshold<=1 / k==0 are rejected before any device work
    (derandomize.rs:274-276, translate.rs:268-270)."""
    for fn, arg in ((derandomize.derandomize_ms_vec, [1, 2]), (translate.translate_ms_vec, [1, 2])):
        with pytest.raises(kbo_amd.KboError) as e:
            fn(arg, 3, 2)
        assert e.value.code == -2
        with pytest.raises(kbo_amd.KboError) as e:
            fn([1, 2, 3], 3, 1)
        assert e.value.code == -3
        with pytest.raises(kbo_amd.KboError) as e:
            fn([1, 2, 3], 0, 2)
        assert e.value.code == -4
    with pytest.raises(kbo_amd.KboError) as e:  # derandomize.rs:229 curr_noisy_ms <= k
        derandomize.derandomize_ms_vec([1, 9, 3], 3, 2)
    assert e.value.code == -9


def test_empty_query_and_map_opts_checks():
    sbwt, lcs = kbo_amd.build([b"ACGTACGTTGCA"], kbo_amd.BuildOpts(k=4))
    with pytest.raises(kbo_amd.KboError) as e:  # index.rs:248
        kbo_amd.index.query_sbwt(b"", sbwt, lcs)
    assert e.value.code == -1
    with pytest.raises(kbo_amd.KboError) as e:  # lib.rs:729
        kbo_amd.map(b"ACGTACGT", sbwt, lcs, kbo_amd.MapOpts())
    assert e.value.code == -6
    with pytest.raises(kbo_amd.KboError) as e:  # lib.rs:559
        kbo_amd.call(sbwt, lcs, b"ACGTACGTACGTAGCTAGCTAGCATCGATCGACTAGCTAC", kbo_amd.CallOpts())
    assert e.value.code == -6


def test_index_opts_round_trip_and_checks(tmp_path):
    """kbo_index_opts_t needs no GPU until a batch runs: defaults inherit everything, set / get round trip, the struct's size and
    the device count are checked"""
    L = kbo_amd.lib()
    o = _capi.IndexOpts()
    kbo_amd.check(L.kbo_index_opts_default(C.byref(o)))
    assert o.struct_size == C.sizeof(_capi.IndexOpts) and o.plan == o.depth_table == o.depth_table_anchors == _capi.OPT_INHERIT
    assert o.slab_bytes == 0 and o.n_devices == -1
    sbwt, _ = kbo_amd.build([b"ACGTTGCATGCATGCAAGTCGATCGATTTGACCATG" * 3], kbo_amd.BuildOpts(k=7))
    assert sbwt.get_opts() == {"plan": _capi.OPT_INHERIT, "depth_table": _capi.OPT_INHERIT, "depth_table_anchors": _capi.OPT_INHERIT,
                               "slab_bytes": 0, "devices": None}
    sbwt.set_opts(plan=0, depth_table=99, depth_table_anchors=5, slab_bytes=1, devices=[])
    assert sbwt.get_opts() == {"plan": 0, "depth_table": 17, "depth_table_anchors": 1, "slab_bytes": 1 << 16, "devices": []}
    sbwt.set_opts(depth_table=-5)
    assert sbwt.get_opts()["depth_table"] == -1 and sbwt.get_opts()["plan"] == 0
    o.struct_size = 12
    assert L.kbo_index_set_opts(sbwt._h, C.byref(o)) == -4
    kbo_amd.check(L.kbo_index_opts_default(C.byref(o)))
    o.n_devices = 17
    assert L.kbo_index_set_opts(sbwt._h, C.byref(o)) == -4
    assert L.kbo_index_set_opts(None, C.byref(o)) == -4 and L.kbo_index_get_opts(sbwt._h, None) == -4


def test_compute_fails_loudly_without_gpu():
    """No CPU fallback: on a machine without a HIP device every compute call is an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sbwt, lcs = kbo_amd.build([b"ACGTACGTTGCA"], kbo_amd.BuildOpts(k=4))
    for call in (lambda: kbo_amd.index.query_sbwt(b"ACGTAC", sbwt, lcs),
                 lambda: kbo_amd.matches(b"ACGTAC", sbwt, lcs),
                 lambda: derandomize.derandomize_ms_vec([1, 2, 3], 3, 2),
                 lambda: translate.translate_ms_vec([1, 2, 3], 3, 2)):
        with pytest.raises(kbo_amd.KboError) as e:
            call()
        assert e.value.code == -7


def test_from_parts_rejects_inconsistent_indexes(tmp_path):
    """kbo_index_from_parts / kbo_index_load validate what the kernels trust: C[] against the edge bits, LCS < k,
    file length (an index that came from elsewhere must not be able to send the walk out of bounds)."""
    import numpy as np
    from kbo_amd import index as kindex
    sbwt, _ = kbo_amd.build([b"AAAGAACCA-TCAGGGCG"], kbo_amd.BuildOpts(k=3))
    rows, Carr, lcs = sbwt.export_parts()
    ok = kindex.SbwtIndexVariant.from_parts(3, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
    assert ok.n_sets() == 16
    for breakage in ("C", "lcs", "bits", "k"):
        r2, C2, l2, k2 = [r.copy() for r in rows], list(Carr), lcs.copy(), 3
        if breakage == "C":
            C2[2] += 1
        elif breakage == "lcs":
            l2[5] = 3
        elif breakage == "bits":
            r2[0][0] |= np.uint64(1 << 2)
        else:
            k2 = 0
        with pytest.raises(AssertionError):
            kindex.SbwtIndexVariant.from_parts(k2, sbwt.n_sets(), sbwt.n_kmers(), r2, C2, l2)
    # the single-file cache format
    path = str(tmp_path / "idx.kbohip")
    kindex.save_flat(path, sbwt)
    raw = open(path, "rb").read()
    open(path, "wb").write(raw + b"x")          # trailing bytes
    with pytest.raises(AssertionError):
        kindex.load_flat(path)
    open(path, "wb").write(raw[:-3])             # truncated
    with pytest.raises(AssertionError):
        kindex.load_flat(path)
    open(path, "wb").write(raw)
    assert kindex.load_flat(path)[0].n_sets() == 16
    # the file carries the path cover of the plan-guided walk (text n + pos 4 n + node_at 4 n bytes behind a tag); a loaded
    # index hands out the same cover without computing it again, a cover that claims an edge the subset matrix does not
    # have is refused
    n = 16
    assert raw[-(9 * n + 8):-(9 * n)] == b"KBOPCOV1"
    loaded, _ = kindex.load_flat(path)
    assert all((a == b).all() for a, b in zip(sbwt.path_cover(), loaded.path_cover()))
    text_off = len(raw) - 9 * n
    for p in range(n):
        if raw[text_off + p] not in (0, ord("T")):
            broken = bytearray(raw)
            broken[text_off + p] = ord("T")  # a label the edge does not carry
            open(path, "wb").write(bytes(broken))
            with pytest.raises(AssertionError):
                kindex.load_flat(path)
            break
    broken = bytearray(raw)
    broken[len(raw) - 4 * n:len(raw) - 4 * n + 4] = broken[len(raw) - 4 * n + 4:len(raw) - 4 * n + 8]  # node_at no permutation
    open(path, "wb").write(bytes(broken))
    with pytest.raises(AssertionError):
        kindex.load_flat(path)
    open(path, "wb").write(raw[:8 + 56 + 4 * 8 + n])  # an index file without a cover (what earlier versions wrote) still loads
    assert kindex.load_flat(path)[0].n_sets() == 16


def test_sbwt_lcs_file_pair_round_trip_and_foreign_payload(tmp_path):
    """index::serialize_sbwt / load_sbwt (index.rs:128-151, 195-212; the reference's own test is a round trip,
    index.rs:277-296).  The pair is written under its own names (<prefix>.sbwt.kbohip / .lcs.kbohip) so that kbo-cli
    never takes it for a crate-written index; the header the reference writes is reproduced byte for byte; a
    crate-written <prefix>.sbwt is refused as unsupported, not guessed at."""
    import struct
    from kbo_amd import index as kindex
    sbwt, lcs = kbo_amd.build([b"AAAGAACCA-TCAGGGCG"], kbo_amd.BuildOpts(k=3))
    prefix = str(tmp_path / "serialized_index_1")
    kindex.serialize_sbwt(prefix, sbwt, lcs)
    assert not os.path.exists(prefix + ".sbwt") and not os.path.exists(prefix + ".lcs")  # the upstream names stay free
    raw = open(prefix + ".sbwt.kbohip", "rb").read()
    assert raw[:20] == struct.pack("<Q", 12) + b"SubsetMatrix"          # index.rs:139-140
    loaded, _ = kindex.load_sbwt(prefix)
    a, b = sbwt.export_parts(), loaded.export_parts()
    assert (loaded.k(), loaded.n_sets(), loaded.n_kmers()) == (3, 16, 13)
    assert all((x == y).all() for x, y in zip(a[0], b[0])) and a[1] == b[1] and (a[2] == b[2]).all()
    # a crate-written pair under the upstream names (same header, the sbwt crate's payload behind it) and no pair of ours
    other = str(tmp_path / "from_kbo_cli")
    open(other + ".sbwt", "wb").write(raw[:20] + b"\x10" + bytes(200))
    open(other + ".lcs", "wb").write(bytes(64))
    with pytest.raises(kbo_amd.KboError) as e:
        kindex.load_sbwt(other)
    assert e.value.code == -8  # KBO_E_UNSUPPORTED
    # a pair this library wrote under the upstream names before they were moved aside is still read
    legacy = str(tmp_path / "legacy")
    open(legacy + ".sbwt", "wb").write(raw)
    open(legacy + ".lcs", "wb").write(open(prefix + ".lcs.kbohip", "rb").read())
    assert kindex.load_sbwt(legacy)[0].n_sets() == 16
    open(prefix + ".sbwt.kbohip", "wb").write(b"garbage")
    with pytest.raises(AssertionError):
        kindex.load_sbwt(prefix)
    open(prefix + ".sbwt.kbohip", "wb").write(raw)
    open(prefix + ".lcs.kbohip", "wb").write(open(prefix + ".lcs.kbohip", "rb").read()[:-1])  # LCS file shorter than the index
    with pytest.raises(AssertionError):
        kindex.load_sbwt(prefix)
    with pytest.raises(AssertionError):
        kindex.load_sbwt(str(tmp_path / "no_such_prefix"))


@pytest.mark.parametrize("k,revcomp", [(3, False), (5, True), (31, False), (31, True), (64, False)])
def test_run_automaton_equals_the_walk_of_a_one_sequence_index(oracle, k, revcomp):
    """kbo_call_batch never builds the per-sequence index of lib.rs:553: the depths of the reference-side walk
    (variant_calling.rs:280) come from a suffix automaton of the sequence's ACGT-runs of >= k characters.  Here those depths
    against the oracle's literal walk of a really built one-sequence index: sequences with N's (runs shorter than k have no
    rows), repeats, reverse complements; k-mers cut from the sequence, mutated, '$'-padded, with junk."""
    rng = np.random.default_rng(1000 + k + int(revcomp))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for trial in range(6):
        n = int(rng.integers(2 * k + 5, 3000))
        seq = acgt[rng.integers(0, 4, n)].copy()
        if trial % 2:
            seq[rng.integers(0, n, max(1, n // 60))] = ord("N")        # runs, some shorter than k
        if trial % 3 == 0 and n > 400:
            seq[300:400] = seq[50:150]                                   # a repeat
        if trial == 5:
            seq[:] = acgt[rng.integers(0, 2, n)]                         # low complexity
        ora = oracle.Index.build([seq.tobytes()], k=k, add_revcomp=revcomp)
        kmers = []
        for _ in range(300):
            a = int(rng.integers(0, n - k))
            km = seq[a:a + k].copy()
            kind = int(rng.integers(0, 5))
            if kind == 1:
                km[int(rng.integers(0, k))] = acgt[int(rng.integers(0, 4))]
            elif kind == 2:
                km[:int(rng.integers(1, k))] = ord("$")
            elif kind == 3:
                km = acgt[rng.integers(0, 4, k)]
            elif kind == 4 and revcomp:
                km = (np.frombuffer(bytes(km), dtype=np.uint8)[::-1]).copy()
                km = np.array([{65: 84, 67: 71, 71: 67, 84: 65}.get(int(c), int(c)) for c in km], dtype=np.uint8)
            kmers.append(np.asarray(km, dtype=np.uint8))
        flat = np.concatenate(kmers)
        got = np.zeros(len(kmers) * k, dtype=np.uint32)
        kbo_amd.check(kbo_amd.lib().kbo_run_automaton_depths(seq.ctypes.data, n, k, int(revcomp), flat.ctypes.data, len(kmers),
                                                              got.ctypes.data))
        for x, km in enumerate(kmers):
            d, _, _ = ora.matching_statistics(km.tobytes())
            assert np.array_equal(got[x * k:(x + 1) * k], d.astype(np.uint32)), (k, revcomp, trial, x)


def test_pack_reads_and_unpack_matches_round_trip():
    """The host helpers of the packed entry points (kbo_hip.h): sequence s = ceil(len / 16) u32 words, base i in bits
    2 (i mod 16) of word i / 16; non-ACGT bytes in an ascending side list; M, -, X, R = 0 .. 3 on the way back."""
    from kbo_amd import batch
    rng = np.random.default_rng(77)
    lens = [1, 3, 15, 16, 17, 31, 32, 33, 150, 151, 1000] + [int(x) for x in rng.integers(1, 400, 200)]
    seqs = []
    for n in lens:
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
        if n > 5 and rng.random() < 0.4:
            a[rng.integers(0, n, 2)] = rng.choice(list(b"Nn$x-"))
        seqs.append(a)
    concat = np.concatenate(seqs)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    words, pos, byt = batch.pack_reads(concat, offsets)
    assert len(words) == sum((n + 15) // 16 for n in lens)
    assert np.all(np.diff(pos.astype(np.int64)) > 0)
    back = np.zeros(len(concat), dtype=np.uint8)
    w = 0
    for s, n in enumerate(lens):  # unpack by the layout's definition
        for i in range(n):
            back[int(offsets[s]) + i] = b"ACGT"[(int(words[w + i // 16]) >> (2 * (i % 16))) & 3]
        w += (n + 15) // 16
    back[pos.astype(np.int64)] = byt
    assert np.array_equal(back, concat)
    # output side: pack M - X R by the definition, unpack with the helper
    chars = np.frombuffer(b"M-XR", dtype=np.uint8)[rng.integers(0, 4, len(concat))]
    code = {77: 0, 45: 1, 88: 2, 82: 3}
    ow = np.zeros(len(words), dtype=np.uint32)
    w = 0
    for s, n in enumerate(lens):
        for i in range(n):
            ow[w + i // 16] |= np.uint32(code[int(chars[int(offsets[s]) + i])] << (2 * (i % 16)))
        w += (n + 15) // 16
    assert np.array_equal(batch.unpack_matches(ow, offsets), chars)


@pytest.mark.parametrize("revcomp", [False, True])
def test_sharded_index_counts_the_union(oracle, revcomp):
    """An index whose rows would not fit 32-bit row numbers is built as shards (groups of sequences, strands apart); forced
    here on a small input.  n_kmers - what the derandomisation threshold needs - must be the number of distinct k-mers of
    the union, i.e. what the one index over everything has; what needs rows of that one index is refused."""
    from kbo_amd import index as kindex
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [acgt[rng.integers(0, 4, int(n))].tobytes() for n in rng.integers(200, 3000, 9)]
    seqs[3] = seqs[0][100:900] + b"N" + seqs[5][:500]     # shared k-mers across groups
    seqs[7] = bytes(acgt[3 - np.searchsorted(acgt, np.frombuffer(seqs[1], dtype=np.uint8))][::-1])  # the reverse complement of another
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    probe = (seqs[2][50:400] + seqs[6][::-1].translate(comp)[:300] + b"N" + seqs[3][780:830] + seqs[8][-200:]
             + bytes(acgt[rng.integers(0, 4, 300)]))
    L = kbo_amd.lib()
    for k in (5, 31):
        ora = oracle.Index.build(seqs, k=k, add_revcomp=revcomp)
        for shards in (2, 3, 7):
            try:
                L.kbo_set_index_shards(shards)
                sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, add_revcomp=revcomp, num_threads=2))
            finally:
                L.kbo_set_index_shards(0)
            assert sbwt.shards() >= min(shards, len(seqs))
            assert sbwt.k() == k and sbwt.n_kmers() == ora.n_kmers, (k, shards, revcomp)
            # the shards are ordinary indexes over disjoint parts of the input (rows add up), and the depth of the walk
            # against the index of everything is the maximum of the depths against them - the oracle on both sides
            parts = [sbwt.shard(i) for i in range(sbwt.shards())]
            assert sum(p.n_sets() for p in parts) == sbwt.n_sets() and all(p.shards() == 1 for p in parts)
            with pytest.raises(IndexError):
                sbwt.shard(sbwt.shards())
            d_parts = []
            for p in parts:
                rows, Carr, lcs = p.export_parts()
                d_parts.append(oracle.Index.from_parts(k, p.n_sets(), p.n_kmers(), rows, Carr, lcs).matching_statistics(probe)[0])
            assert np.array_equal(np.max(d_parts, axis=0), ora.matching_statistics(probe)[0]), (k, shards, revcomp)
            for refused in (sbwt.export_parts, sbwt.path_cover, lambda: kindex.save_flat("/tmp/never_written.kbohip", sbwt)):
                with pytest.raises(kbo_amd.KboError) as e:
                    refused()
                assert e.value.code == -8  # KBO_E_UNSUPPORTED
    one, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=31))
    assert one.shards() == 1
