"""kbo_map_batch_dev's one-kernel route (kbo_amd/csrc/map_kernels.hip: MS -> derandomize -> translate [-> relative_to_ref]
for 64 reads per wave, 2-bit packed) against the CPU oracle, every base: read shapes (ragged lengths down to 3 bases, reads
shorter than a seed), contents (substitutions up to 8 %, indels, junk heads, N and lower-case bytes, unrelated reads, reads of
the other strand), index shapes (several contigs with repeats: many paths in the cover; k from 11 to 63), both stretch widths
of the kernel (tables of up to 15 bases / of 16 and 17), with and without formatting, with and without the MS values."""
import numpy as np
import pytest

import kbo_amd
from kbo_amd import batch, synth
from gpu_helpers import threads

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate(rng, reads, sub=0.0, indel=0.0, junk=0, n_rate=0.0, lower=0.0):
    """list of uint8 arrays -> list of uint8 arrays"""
    out = []
    for r in reads:
        r = r.copy()
        if sub:
            hit = rng.random(len(r)) < sub
            r[hit] = ACGT[rng.integers(0, 4, int(hit.sum()))]
        if junk:
            r[:junk] = ACGT[rng.integers(0, 4, min(junk, len(r)))]
        if indel and len(r) > 40 and rng.random() < indel:
            p = int(rng.integers(10, len(r) - 10))
            if rng.random() < 0.5:
                r = np.concatenate([r[:p], r[p + int(rng.integers(1, 4)):]])
            else:
                r = np.concatenate([r[:p], ACGT[rng.integers(0, 4, int(rng.integers(1, 4)))], r[p:]])
        if n_rate and rng.random() < n_rate:
            r[int(rng.integers(0, len(r)))] = ord("N")
        if lower and rng.random() < lower:
            p = int(rng.integers(0, len(r)))
            r[p] = r[p] | 0x20
        out.append(r)
    return out


def _batch_of(reads):
    concat = np.concatenate(reads)
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    return concat, offsets


def _check(oracle, ora, sbwt, concat, offsets, p=1e-7, expect_fused=True, expect_fused_without_ms=None):
    import torch
    exp_chars, exp_d = ora.matches_batch(concat, offsets, p, n_threads=threads(), want_d=True)
    exp_map = np.frombuffer(oracle.relative_to_ref(concat, exp_chars), dtype=np.uint8)
    for fmt, want_ms in ((False, True), (True, False), (False, False)):  # (without the MS values: the kernel's direct form)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"), max_error_prob=p, format=fmt, want_ms=want_ms)
        dev.ms.fill_(0xEE)
        dev.chars.fill_(0xEE)
        kbo_amd.lib().kbo_set_plan(1, 0, 0)  # (a batch that gave the plan up - 8 % substitutions - holds the copy off: clear it)
        dev.run()
        torch.cuda.synchronize()
        assert dev.fused == (expect_fused if want_ms or expect_fused_without_ms is None else expect_fused_without_ms)
        got = dev.chars[:dev.total].cpu().numpy()
        want = exp_map if fmt else exp_chars
        if not np.array_equal(got, want):
            bad = np.flatnonzero(got != want)
            s = int(np.searchsorted(offsets, bad[0], side="right")) - 1
            a, b = int(offsets[s]), int(offsets[s + 1])
            raise AssertionError("format %s, MS wanted %s: read %d (len %d) of %d: first bad base %d, %d bad bases in %d reads\n got  %s\n want %s\n read %s\n ms   %s" % (
                fmt, want_ms, s, b - a, len(offsets) - 1, int(bad[0]) - a, len(bad), len(np.unique(np.searchsorted(offsets, bad, side="right"))),
                got[a:b].tobytes(), want[a:b].tobytes(), concat[a:b].tobytes(), [int(v) for v in exp_d[a:b]]))
        if want_ms:
            assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d)
        del dev


@pytest.mark.parametrize("k", [31, 11, 19, 51, 63])
def test_read_shapes_and_contents(oracle, k):
    rng = np.random.default_rng(100 + k)
    g = synth.genome(300_000, seed=900 + k)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=k, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=k)
    sbwt.to_device(-1)
    assert sbwt.depth_table_order() > 0

    def take(n, lo, hi):
        lens = rng.integers(lo, hi + 1, n)
        starts = rng.integers(0, len(g) - 200, n)
        return [g[a:a + l] for a, l in zip(starts, lens)]
    other = synth.genome(50_000, seed=77)
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    sets = [
        _mutate(rng, take(5000, 150, 150), sub=0.01),
        _mutate(rng, take(3000, 150, 150)),                              # error-free
        _mutate(rng, take(3000, 150, 150), sub=0.05),
        _mutate(rng, take(2000, 150, 150), sub=0.08),                    # most reads overflow their mismatch list
        _mutate(rng, take(4000, 3, 160), sub=0.01),                      # ragged, down to three bases
        _mutate(rng, take(2000, 3, 30), sub=0.02),                       # shorter than a seed
        _mutate(rng, take(3000, 100, 157), sub=0.01, indel=0.5),
        _mutate(rng, take(3000, 150, 150), sub=0.01, junk=12),           # the first seed fails
        _mutate(rng, take(3000, 140, 160), sub=0.01, n_rate=0.2, lower=0.1),
        [other[a:a + 150] for a in rng.integers(0, len(other) - 150, 2000)],            # unrelated
        [comp[g[a:a + 150]][::-1].copy() for a in rng.integers(0, len(g) - 150, 2000)],  # the other strand
        _mutate(rng, take(3000, 150, 150), sub=0.01, junk=70),           # no seed in the first 64 bases: seeded from the back
        [np.concatenate([a[:int(c)], b[int(c):]]) for a, b, c in zip(take(3000, 150, 150), take(3000, 150, 150), rng.integers(20, 131, 3000))],  # chimeras
        _mutate(rng, _mutate(rng, take(2000, 120, 154), indel=1.0), sub=0.01, indel=1.0),  # two insertions / deletions
        _mutate(rng, take(777, 128, 128), sub=0.01),                     # lengths that are multiples of 16 / 32
        _mutate(rng, take(63, 160, 160), sub=0.01),                      # less than one wave
    ]
    for reads in sets:
        _check(oracle, ora, sbwt, *_batch_of(reads))
    mixed = [r for reads in sets for r in reads[:300]]
    order = rng.permutation(len(mixed))
    _check(oracle, ora, sbwt, *_batch_of([mixed[i] for i in order]))
    _check(oracle, ora, sbwt, *_batch_of([mixed[i] for i in order][:500]), p=0.01)  # a low threshold: anchors and 'R's


@pytest.mark.parametrize("k", [31, 21])
def test_insertions_and_deletions_near_either_end(oracle, k):
    """the second diagonal found beside the first one (the read's first / last 16 or 8 bases against the text one to three bases
    off: map_kernels.hip 2b) - breaks 1 .. 40 bases from either end, with substitutions on top, lengths from 28 up, and a genome
    with short-period tandem arrays, where several shifts fit the same bases; every read against the oracle.
    Semantics: /root/reference/src/translate.rs:188-206 (what an insertion / a deletion looks like in the characters)"""
    rng = np.random.default_rng(4100 + k)
    g = synth.genome(400_000, seed=4200 + k)
    for _ in range(300):  # tandem arrays of period 1 - 4, 10 - 60 bases long
        a = int(rng.integers(0, len(g) - 100))
        unit = ACGT[rng.integers(0, 4, int(rng.integers(1, 5)))]
        n = int(rng.integers(10, 61))
        g[a:a + n] = np.tile(unit, n // len(unit) + 1)[:n]
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=k, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=k)
    sbwt.to_device(-1)
    reads = []
    for i in range(24_000):
        L = int(rng.integers(28, 158))
        a = int(rng.integers(0, len(g) - 200))
        r = g[a:a + L + 4].copy()
        d = int(rng.integers(1, min(41, L - 1)))
        pos = d if i % 2 else L - d  # from the front / from the back
        z = int(rng.integers(1, 5))   # (four bases: beyond what the search beside the diagonal covers)
        if rng.random() < 0.5:
            r = np.concatenate([r[:pos], r[pos + z:]])
        else:
            r = np.concatenate([r[:pos], ACGT[rng.integers(0, 4, z)], r[pos:]])
        r = r[:L]
        hit = rng.random(len(r)) < (0.0, 0.01, 0.03)[i % 3]
        r[hit] = ACGT[rng.integers(0, 4, int(hit.sum()))]
        reads.append(r)
    _check(oracle, ora, sbwt, *_batch_of(reads))


def test_many_paths_and_both_stretch_widths(oracle):
    """several contigs that share repeats (the cover has many paths: path starts inside reads), separated by N; then the same
    reads with a table of 16 and of 17 bases forced onto the small index (the kernel's 18-base stretches, 64-bit keys)."""
    rng = np.random.default_rng(5)
    L = kbo_amd.lib()
    base = synth.genome(60_000, seed=31)
    rep = synth.genome(800, seed=32)
    contigs = []
    for c in range(12):
        a = int(rng.integers(0, 50_000))
        piece = base[a:a + int(rng.integers(3000, 9000))].copy()
        ins = int(rng.integers(100, len(piece) - 900))
        contigs.append(np.concatenate([piece[:ins], rep, piece[ins:]]))
    seqs = [c.tobytes() for c in contigs]
    cat = np.concatenate(contigs)
    k = 31
    ora = oracle.Index.build(seqs, k=k)
    reads = []
    for c in contigs:
        for a in rng.integers(0, len(c) - 160, 400):
            reads.append(c[a:a + int(rng.integers(60, 161))])
    reads = _mutate(rng, reads, sub=0.015)
    reads += [cat[a:a + 150] for a in rng.integers(0, len(cat) - 150, 1500)]  # reads across contig junctions
    concat, offsets = _batch_of(reads)
    for order, anchors in ((0, 0), (16, 0), (17, 0), (9, 0), (9, 1), (10, 1), (12, 1)):
        # (anchors on a table with a thin margin - what indexes of 24 Mi rows and more get: windows that are present are priced by
        # their exact depth off the path-cover text instead of failing the read's proof)
        L.kbo_set_depth_table(order)
        L.kbo_set_depth_table_anchors(anchors)
        sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, num_threads=threads()))
        sbwt.to_device(-1)
        if order:
            assert sbwt.depth_table_order() == order
        assert (sbwt.device_layout()["anchor_bytes"] > 0) == bool(anchors)
        _check(oracle, ora, sbwt, concat, offsets)
    L.kbo_set_depth_table(0)
    L.kbo_set_depth_table_anchors(-1)


def test_c2_shape_the_one_kernel_every_read(oracle):
    """the batch bench.py times (5 Mbp index, 1 M reads of 150 bases, 1 % substitutions), formatted and not, every read"""
    g = synth.genome(5_000_000)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    concat, offsets = synth.reads(g, 1_000_000, 150, 0.01)
    _check(oracle, ora, sbwt, concat, offsets)


def test_how_many_reads_the_kernel_leaves_to_the_second_pass():
    """C2's index (5 Mbp, k = 31): the share of the reads map_reads_kernel flags for the plain walk (kbo_plan_flags_dev) - reads
    with 1 % substitutions (0.37 %; a window that starts at a mismatch and is in the index is settled by the next
    window's entry), and reads with one insertion / deletion of 1 - 3 bases anywhere on top (1.1 %, flat over its position: the
    second diagonal found beside the first).  Loose bounds: what they guard is the mechanism, not the last tenth of a per cent."""
    import torch
    g = synth.genome(5_000_000)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    dev0 = torch.device("cuda:0")
    N, LEN = 100_000, 150
    concat, offsets = synth.reads(g, N, LEN, 0.01)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=False, want_ms=False)
    dev.run()
    torch.cuda.synchronize()
    assert dev.fused
    share = float((dev.plan_flags() != 0).mean())
    assert share < 0.008, share
    rng = np.random.default_rng(3)
    start = rng.integers(8, len(g) - LEN - 8, N)
    pos = rng.integers(1, LEN - 1, N)
    size = rng.integers(1, 4, N)
    ins = rng.random(N) < 0.5
    i = np.arange(LEN)[None, :]
    shift = np.where(i >= pos[:, None], np.where(ins[:, None], -np.minimum(size[:, None], i - pos[:, None]), size[:, None]), 0)
    reads = g[start[:, None] + i + shift]
    new = ins[:, None] & (i >= pos[:, None]) & (i < (pos + size)[:, None])
    reads = np.where(new, ACGT[rng.integers(0, 4, (N, LEN))], reads)
    hit = rng.random((N, LEN)) < 0.01
    reads = np.where(hit, ACGT[(np.searchsorted(ACGT, reads) + rng.integers(1, 4, (N, LEN))) % 4], reads)
    concat = np.ascontiguousarray(reads.reshape(-1))
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=False, want_ms=False)
    dev.run()
    torch.cuda.synchronize()
    fl = dev.plan_flags() != 0
    assert dev.fused and float(fl.mean()) < 0.03, float(fl.mean())
    for lo, hi in ((10, 30), (120, 140)):  # (breaks near either end were 12 - 25 % before the search beside the diagonal)
        sel = (pos >= lo) & (pos < hi)
        assert float(fl[sel].mean()) < 0.06, (lo, hi, float(fl[sel].mean()))


def test_routes_that_are_not_the_one_kernel(oracle):
    """reads longer than 160 bases with the MS values wanted, a copy without a depth table, a held-off copy: kbo_map_batch_dev takes
    the two kernels (longer reads without the MS values: the kernel for sequences of any length, long_kernels.hip)"""
    L = kbo_amd.lib()
    rng = np.random.default_rng(9)
    g = synth.genome(200_000, seed=11)
    ora = oracle.Index.build([g.tobytes()], k=31)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    reads = _mutate(rng, [g[a:a + 300] for a in rng.integers(0, len(g) - 300, 2000)], sub=0.01)
    _check(oracle, ora, sbwt, *_batch_of(reads), expect_fused=False, expect_fused_without_ms=True)
    L.kbo_set_depth_table(-1)
    sb2, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    reads = _mutate(rng, [g[a:a + 150] for a in rng.integers(0, len(g) - 150, 2000)], sub=0.01)
    _check(oracle, ora, sb2, *_batch_of(reads), expect_fused=False)
    L.kbo_set_depth_table(0)


def test_two_batches_in_flight(oracle):
    """kbo_map_batch_dev_tail: the kernel on one stream, the second pass on another, beside the next batch's kernel; a batch's
    buffers are used again behind its own second pass.  Three batches of different shapes in turn, several rounds, then every read."""
    import torch
    rng = np.random.default_rng(21)
    g = synth.genome(400_000, seed=41)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    dev0 = torch.device("cuda:0")
    S, T = torch.cuda.Stream(dev0), torch.cuda.Stream(dev0)
    kbo_amd.lib().kbo_set_plan(1, 0, 0)
    sets = []
    for n, sub, fmt in ((40_000, 0.01, True), (25_000, 0.02, False), (33_333, 0.0, True)):
        reads = _mutate(rng, [g[a:a + int(l)] for a, l in zip(rng.integers(0, len(g) - 200, n), rng.integers(30, 158, n))], sub=sub, indel=0.2)  # (an insertion adds up to 3 bases: 160 at most)
        concat, offsets = _batch_of(reads)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=fmt, want_ms=False)
        exp_chars = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads())
        want = np.frombuffer(oracle.relative_to_ref(concat, exp_chars), dtype=np.uint8) if fmt else exp_chars
        sets.append((dev, want, torch.cuda.Event()))
    with torch.cuda.stream(S):
        for rnd in range(4):
            for dev, _, done in sets:
                if rnd:
                    S.wait_event(done)
                    dev.chars.fill_(0xEE)  # (on S, behind the batch's last second pass)
                kbo_amd.lib().kbo_set_plan(1, 0, 0)  # (no hold-off behind the batch with 2 % substitutions)
                dev.run(S, tail_stream=T)
                assert dev.fused
                done.record(T)
    torch.cuda.synchronize()
    for dev, want, _ in sets:
        assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), want)
    # tail_stream == stream is kbo_map_batch_dev
    dev, want, _ = sets[0]
    dev.chars.fill_(0xEE)
    dev.run(S, tail_stream=S)
    torch.cuda.synchronize()
    assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), want)


def test_packed_native_kernel_device_resident(oracle):
    """kbo_matches_packed_dev: the reads as 2-bit words into the kernel, the characters as 2-bit words out of it - ragged and equally
    long reads, reads with bytes that are no bases (the side list), indels, reads the kernel leaves to the second pass, with and
    without a tail stream - against the oracle, every read; and the host entry point kbo_matches_batch_packed over the same reads."""
    import torch
    rng = np.random.default_rng(33)
    g = synth.genome(500_000, seed=43)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    dev0 = torch.device("cuda:0")
    T = torch.cuda.Stream(dev0)

    def take(n, lo, hi):
        return [g[a:a + int(l)] for a, l in zip(rng.integers(0, len(g) - 200, n), rng.integers(lo, hi + 1, n))]
    sets = [
        _mutate(rng, take(30_000, 150, 150), sub=0.01),                                   # equally long: words per read known
        _mutate(rng, take(20_000, 3, 157), sub=0.02, indel=0.3),                          # ragged, down to three bases
        _mutate(rng, take(10_000, 100, 160), sub=0.01, n_rate=0.3, lower=0.2),            # the side list
        _mutate(rng, take(5_000, 128, 128), sub=0.06),                                    # multiples of 16; many reads left to the second pass
        _mutate(rng, take(63, 17, 160), sub=0.01),
    ]
    for reads in sets:
        concat, offsets = _batch_of(reads)
        want = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads())
        kbo_amd.lib().kbo_set_plan(1, 0, 0)
        pb = batch.PackedDeviceBatch(sbwt, concat, offsets, device=dev0)
        for tail in (None, T):
            pb.words_out.fill_(-1)
            pb.run(tail_stream=tail)
            torch.cuda.synchronize()
            got = pb.chars()
            if not np.array_equal(got, want):
                bad = np.flatnonzero(got != want)
                s = int(np.searchsorted(offsets, bad[0], side="right")) - 1
                a, b = int(offsets[s]), int(offsets[s + 1])
                raise AssertionError("read %d (len %d): %d bad bases\n got  %s\n want %s" % (s, b - a, len(bad), got[a:b].tobytes(), want[a:b].tobytes()))
        # the bits behind a read's last base inside its last word are zero, as from the byte route + pack
        words, pos, byt = batch.pack_reads(concat, offsets)
        host = batch.matches_batch_packed(sbwt, words, offsets, pos, byt)
        assert np.array_equal(host, pb.words_out[:len(words)].cpu().numpy().view(np.uint32))
        assert np.array_equal(batch.unpack_matches(host, offsets), want)


def test_find_on_the_device_counts_runs_in_the_kernel(oracle):
    """kbo_find_batch_dev: the characters and format::run_lengths_gapped of them for a device-resident batch.  With max_gap_len = 0 the
    one kernel counts the runs of the reads it finishes (and the second pass those of the reads left to it), so the
    records come from one pass over the characters; with a gap length the usual two.  Against the oracle's literal run lengths, every
    read: ragged reads down to 3 bases, substitutions to 6 %, indels, N's, unrelated reads (no run at all), with and without a tail
    stream; sequences longer than the kernel takes (two kernels, then count + emit)."""
    import torch
    rng = np.random.default_rng(77)
    g = synth.genome(400_000, seed=71)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    dev0 = torch.device("cuda:0")
    T = torch.cuda.Stream(dev0)
    other = synth.genome(30_000, seed=5)

    def take(n, lo, hi):
        return [g[a:a + int(l)] for a, l in zip(rng.integers(0, len(g) - 700, n), rng.integers(lo, hi + 1, n))]
    sets = [
        (_mutate(rng, take(20_000, 150, 150), sub=0.01), True),
        (_mutate(rng, take(15_000, 3, 157), sub=0.03, indel=0.3, n_rate=0.1), True),
        (_mutate(rng, take(5_000, 120, 160), sub=0.06) + [other[a:a + 150] for a in rng.integers(0, len(other) - 150, 2000)], True),
        (_mutate(rng, take(3_000, 100, 470), sub=0.01), True),  # longer than 160 bases: the kernel for sequences of any length
    ]
    for reads, one_kernel in sets:
        concat, offsets = _batch_of(reads)
        exp_chars = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads())
        for gap in (0, 3):
            exp_r, exp_o = oracle.run_lengths_batch(exp_chars, offsets, gap)
            for tail in (None, T):
                kbo_amd.lib().kbo_set_plan(1, 0, 0)
                dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=False, want_ms=False)
                dev.run_find(gap, tail_stream=tail, runs_per_seq=4)
                torch.cuda.synchronize()
                assert dev.fused == one_kernel
                assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars)
                recs, first = dev.run_lengths_host()
                assert np.array_equal(np.asarray(first, dtype=np.uint64), exp_o), (gap, tail is not None)
                assert np.array_equal(np.asarray(recs, dtype=np.uint64).reshape(-1, 7), exp_r), (gap, tail is not None)


def test_the_reads_the_kernel_leaves_in_each_of_the_finishing_kernels_shapes(oracle):
    """finish_reads_kernel (map_kernels.hip) takes the reads map_reads_kernel lists one, four or sixteen a wave by the list's length (up to
    1024 / up to 4096 / beyond): three batches whose lists fall into the three ranges - ragged reads with N's and lower-case bytes among
    them -, every character (formatted and not) and, through the MS-emitting form, every matching statistic against the oracle"""
    import torch
    rng = np.random.default_rng(606)
    g = synth.genome(600_000, seed=66)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    dev0 = torch.device("cuda:0")

    def take(n, lo, hi):
        return [g[a:a + int(l)] for a, l in zip(rng.integers(0, len(g) - 200, n), rng.integers(lo, hi + 1, n))]
    seen = []
    for n, sub, n_rate in ((30_000, 0.01, 0.002), (60_000, 0.05, 0.01), (60_000, 0.09, 0.05)):
        concat, offsets = _batch_of(_mutate(rng, take(n, 3, 160), sub=sub, n_rate=n_rate, lower=n_rate))
        kbo_amd.lib().kbo_set_plan(1, 0, 0)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=False, want_ms=False)
        dev.run()
        torch.cuda.synchronize()
        assert dev.fused
        seen.append(int((dev.plan_flags() != 0).sum()))
        del dev
        _check(oracle, ora, sbwt, concat, offsets)
    assert 0 < seen[0] <= 1024 < seen[1] <= 4096 < seen[2], seen


def test_a_batch_that_gives_the_plan_up_holds_the_copy_off(oracle):
    """20 % substitutions through the one kernel (15 % of the bases differ: 22 mismatches a read, its list holds 13): most reads are left to the second pass, whose kernel (finish_reads_kernel) gives the plan up and says
    so in the copy's pinned word; the next launches over that copy take the two kernels (exact either way) until kbo_set_plan(1, ..)."""
    import torch
    rng = np.random.default_rng(3)
    g = synth.genome(300_000, seed=19)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    dev0 = torch.device("cuda:0")
    bad = _batch_of(_mutate(rng, [g[a:a + 150] for a in rng.integers(0, len(g) - 150, 20_000)], sub=0.20))
    good = _batch_of(_mutate(rng, [g[a:a + 150] for a in rng.integers(0, len(g) - 150, 20_000)], sub=0.01))
    want_bad = ora.matches_batch(*bad, 1e-7, n_threads=threads())
    want_good = ora.matches_batch(*good, 1e-7, n_threads=threads())
    kbo_amd.lib().kbo_set_plan(1, 0, 0)
    d_bad = batch.DeviceBatch(sbwt, *bad, device=dev0, format=False, want_ms=False)
    d_good = batch.DeviceBatch(sbwt, *good, device=dev0, format=False, want_ms=False)
    d_bad.run()
    torch.cuda.synchronize()
    assert d_bad.fused and np.array_equal(d_bad.chars[:d_bad.total].cpu().numpy(), want_bad)
    d_good.run()  # (the copy is held off now)
    torch.cuda.synchronize()
    assert not d_good.fused and np.array_equal(d_good.chars[:d_good.total].cpu().numpy(), want_good)
    kbo_amd.lib().kbo_set_plan(1, 0, 0)
    d_good.run()
    torch.cuda.synchronize()
    assert d_good.fused and np.array_equal(d_good.chars[:d_good.total].cpu().numpy(), want_good)


def test_ms_values_alone_through_the_one_kernel(oracle):
    """kbo_ms_batch_dev over reads and a copy with a depth table: map_reads_kernel in its MS-emitting form, stopped behind the values (no
    characters are made), the plain walk for the reads it leaves.  Every MS byte against the oracle's (index.rs:243-256), next to the
    plan-guided walk (kbo_set_ms_one_kernel(0)) over the same batch; intervals asked for: the walk, as before."""
    import torch
    rng = np.random.default_rng(91)
    g = synth.genome(400_000, seed=17)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    dev0 = torch.device("cuda:0")
    L = kbo_amd.lib()
    other = synth.genome(30_000, seed=6)

    def take(n, lo, hi):
        return [g[a:a + int(l)] for a, l in zip(rng.integers(0, len(g) - 200, n), rng.integers(lo, hi + 1, n))]
    sets = [_mutate(rng, take(20_000, 150, 150), sub=0.01), _mutate(rng, take(8_000, 3, 160), sub=0.04, indel=0.3, n_rate=0.1),
            _mutate(rng, take(3_000, 150, 150), sub=0.08) + [other[a:a + 150] for a in rng.integers(0, len(other) - 150, 2000)]]
    for reads in sets:
        concat, offsets = _batch_of(reads)
        _, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads(), want_d=True)
        for one_kernel in (1, 0):
            L.kbo_set_ms_one_kernel(one_kernel)
            dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=False, want_ms=True)
            dev.ms.fill_(0xEE)
            dev.walk()
            torch.cuda.synchronize()
            assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d), one_kernel
            if one_kernel:  # (the kernel ran: its flags are in the work buffer - a few per cent of such reads at most)
                fl = dev.plan_flags()
                assert 0 < int(np.count_nonzero(fl)) < len(fl) // 2
            del dev
        L.kbo_set_ms_one_kernel(1)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=False, want_ms=True, want_intervals=True)
        dev.walk()
        torch.cuda.synchronize()
        assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d)
        del dev
