"""kbo_map_batch_dev for sequences of ANY length (kbo_amd/csrc/long_kernels.hip: one wave per piece of a sequence - stretches on
diagonals of the text in two bit planes, characters and proof as functions of the planes; flagged pieces by the plain walk + the
literal recurrences) against the CPU oracle, every base: 161 bases ... 1 Mbp, clean / 1 % / 5 % substitutions, insertions and
deletions (ONT-like: every ~80 bases; large ones), N runs, lower case, joins of two places of the genome, unrelated stretches,
sequences of the other strand, several contigs with repeats, reads mixed in, k = 19 ... 63, with and without relative_to_ref;
find's run lengths behind it (lib.rs:612-628, 720-761, 808-821)."""
import numpy as np
import pytest

import kbo_amd
from kbo_amd import batch, synth
from gpu_helpers import threads

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate(rng, src, sub=0.0, indel=0.0, big=0.0):
    """substitutions; insertions / deletions of 1 - 3 bases (of 4 - 60 with probability `big`) starting at `indel` of the bases"""
    n = len(src)
    r = rng.random(n)
    out = []
    i = 0
    ev = np.flatnonzero(r < sub + indel)
    last = 0
    for p in ev:
        if p < last:
            continue
        out.append(src[last:p])
        if r[p] < sub:
            out.append(np.array([ACGT[(int(np.searchsorted(ACGT, src[p])) + int(rng.integers(1, 4))) % 4]], dtype=np.uint8))
            last = p + 1
        else:
            m = int(rng.integers(1, 4)) if rng.random() >= big else int(rng.integers(4, 61))
            if rng.random() < 0.5:
                last = p + m  # deletion
            else:
                out.append(ACGT[rng.integers(0, 4, m)])
                last = p
                out.append(src[p:p + 1])
                last = p + 1
    out.append(src[last:])
    return np.concatenate(out) if out else src.copy()


def _batch_of(seqs):
    concat = np.concatenate(seqs)
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in seqs])]).astype(np.uint64)
    return concat, offsets


def _check(oracle, ora, sbwt, concat, offsets, p=1e-7, expect_long=True, formats=(False, True)):
    import torch
    exp_chars = ora.matches_batch(concat, offsets, p, n_threads=threads())
    exp_map = np.frombuffer(oracle.relative_to_ref(concat, exp_chars), dtype=np.uint8)
    stats = None
    for fmt in formats:
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"), max_error_prob=p, format=fmt, want_ms=False)
        dev.chars.fill_(0xEE)
        dev.run()
        torch.cuda.synchronize()
        assert dev.fused == expect_long
        got = dev.chars[:dev.total].cpu().numpy()
        want = exp_map if fmt else exp_chars
        # (sequences of fewer than 3 bases are left unwritten: derandomize.rs:274-276 asserts on them)
        lens = np.diff(offsets.astype(np.int64))
        keep = np.repeat(lens >= 3, lens)
        if not np.array_equal(got[keep], want[keep]):
            bad = np.flatnonzero((got != want) & keep)
            s = int(np.searchsorted(offsets, bad[0], side="right")) - 1
            a = int(offsets[s])
            q = int(bad[0])
            lo, hi = max(a, q - 60), min(int(offsets[s + 1]), q + 60)
            raise AssertionError("format %s: sequence %d (len %d) of %d: first bad base %d, %d bad bases\n got  %s\n want %s\n seq  %s" % (
                fmt, s, int(offsets[s + 1]) - a, len(offsets) - 1, q - a, len(bad), got[lo:hi].tobytes(), want[lo:hi].tobytes(), concat[lo:hi].tobytes()))
        if expect_long:
            stats = dev.long_stats()
        del dev
    return stats


def _genome(rng, n, contigs=1, repeats=True):
    out = []
    for c in range(contigs):
        g = synth.genome(n // contigs, seed=int(rng.integers(1, 1 << 30)))
        if repeats:
            a, b = int(rng.integers(0, len(g) - 400)), int(rng.integers(0, len(g) - 400))
            g[b:b + 300] = g[a:a + 300]
            t0 = int(rng.integers(0, len(g) - 500))
            g[t0:t0 + 240] = np.tile(g[t0:t0 + 12], 20)
            h0 = int(rng.integers(0, len(g) - 100))
            g[h0:h0 + 40] = ord("A")
        out.append(g)
    return out


def _sequences(rng, contigs, n_seqs, lengths, sub, indel, big=0.0, spice=True):
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    seqs = []
    for s in range(n_seqs):
        src = contigs[int(rng.integers(0, len(contigs)))]
        L = int(lengths[int(rng.integers(0, len(lengths)))])
        L = min(L, len(src) - 1)
        a = int(rng.integers(0, len(src) - L))
        q = _mutate(rng, src[a:a + L], sub, indel, big)
        kind = rng.random() if spice else 1.0
        if kind < 0.08 and len(q) > 10:  # a stretch of something else, or N, or lower case
            p = int(rng.integers(0, len(q)))
            n = int(rng.integers(1, 400))
            what = rng.random()
            q = q.copy()
            if what < 0.4:
                q[p:p + n] = ACGT[rng.integers(0, 4, len(q[p:p + n]))]
            elif what < 0.8:
                q[p:p + n] = ord("N")
            else:
                q[p:p + n] |= 0x20
        elif kind < 0.14:  # a join of two places
            src2 = contigs[int(rng.integers(0, len(contigs)))]
            b2 = int(rng.integers(0, len(src2) - 500))
            q = np.concatenate([q[:len(q) // 2], src2[b2:b2 + 500]])
        elif kind < 0.17:
            q = ACGT[rng.integers(0, 4, L)]  # unrelated
        elif kind < 0.20:
            q = comp[q[::-1]].copy()  # the other strand
        seqs.append(q)
    return seqs


@pytest.mark.parametrize("k", [31, 19, 51, 63])
def test_long_sequences_every_base(oracle, k):
    rng = np.random.default_rng(500 + k)
    contigs = _genome(rng, 400_000, contigs=3)
    sbwt, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=k, num_threads=threads()))
    ora = oracle.Index.build([c.tobytes() for c in contigs], k=k)
    lengths = [161, 200, 700, 961, 962, 1500, 3000, 10_000, 40_000]
    # (the kernel applies where the depth table has fewer bases than the threshold and the threshold is below k: k = 19 over
    # 400 kbp has t = k, those batches take the walk + the derandomize / translate kernels - exact either way)
    t = oracle.random_match_threshold(k, sbwt.n_kmers(), 4, 1e-7)
    sbwt.to_device(-1)
    applies = sbwt.depth_table_order() < t < k
    assert applies == (k != 19)
    for name, sub, indel, big in (("clean", 0.0, 0.0, 0.0), ("1% substitutions", 0.01, 0.0, 0.0), ("5% substitutions", 0.05, 0.0, 0.0),
                                  ("ONT-like", 0.025, 0.0125, 0.0), ("large insertions / deletions", 0.01, 0.002, 0.5)):
        seqs = _sequences(rng, contigs, 60, lengths, sub, indel, big)
        concat, offsets = _batch_of(seqs)
        st = _check(oracle, ora, sbwt, concat, offsets, expect_long=applies)
        print(k, name, st)


def test_every_piece_through_the_second_pass(oracle):
    """kbo_set_map_long(2): every piece is flagged, so every character comes from the plain walk's MS values and the parallel form
    of the literal recurrences (long_derand_kernel: runs of flagged pieces, sequence ends, relative_to_ref)"""
    rng = np.random.default_rng(81)
    contigs = _genome(rng, 300_000, contigs=2)
    sbwt, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([c.tobytes() for c in contigs], k=31)
    kbo_amd.lib().kbo_set_map_long(2)
    for sub, indel, big in ((0.0, 0.0, 0.0), (0.01, 0.0, 0.0), (0.05, 0.0, 0.0), (0.025, 0.0125, 0.0), (0.01, 0.002, 0.5)):
        seqs = _sequences(rng, contigs, 50, [3, 4, 5, 16, 17, 161, 200, 700, 961, 962, 1500, 3000, 10_000], sub, indel, big)
        seqs = [q for q in seqs if len(q) >= 3]
        concat, offsets = _batch_of(seqs)
        st = _check(oracle, ora, sbwt, concat, offsets)
        assert st["flagged"] == st["pieces"]
    kbo_amd.lib().kbo_set_map_long(1)


def test_reads_and_contigs_in_one_batch(oracle):
    """a batch of an assembly's shape: whole contigs, fragments of every length down to 3 bases, reads; sequences of 1 and 2 bases
    among them are left unwritten (derandomize.rs:274-276 asserts on fewer than 3 values)"""
    import torch
    rng = np.random.default_rng(77)
    contigs = _genome(rng, 600_000, contigs=2)
    sbwt, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([c.tobytes() for c in contigs], k=31)
    seqs = _sequences(rng, contigs, 60, [3, 4, 5, 17, 31, 60, 150, 160, 161, 300, 5000], 0.01, 0.001)
    seqs = [q for q in seqs if len(q) >= 3]
    seqs.append(_mutate(rng, contigs[0], 0.01, 0.0005, 0.3))  # a whole contig against its own index, 300 kbp
    seqs.append(contigs[1][1000:200_000].copy())
    concat, offsets = _batch_of(seqs)
    _check(oracle, ora, sbwt, concat, offsets)
    # the same with sequences of 1 and 2 bases in between
    exp = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads())
    mixed, want = [], []
    for i, q in enumerate(seqs):
        mixed.append(q)
        want.append(exp[int(offsets[i]):int(offsets[i + 1])])
        if i % 3 == 0:
            tiny = ACGT[rng.integers(0, 4, 1 + i % 2)]
            mixed.append(tiny)
            want.append(np.full(len(tiny), 0xEE, dtype=np.uint8))
    c2, o2 = _batch_of(mixed)
    dev = batch.DeviceBatch(sbwt, c2, o2, device=torch.device("cuda:0"), want_ms=False)
    dev.chars.fill_(0xEE)
    dev.run()
    torch.cuda.synchronize()
    assert dev.fused
    assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), np.concatenate(want))


def test_one_megabase_sequence(oracle):
    rng = np.random.default_rng(78)
    g = synth.genome(1_200_000, seed=4242)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    q = _mutate(rng, g[50_000:1_050_000], 0.01, 0.001, 0.2)
    concat, offsets = _batch_of([q])
    st = _check(oracle, ora, sbwt, concat, offsets)
    assert st["pieces"] > 1000 and st["flagged"] < st["pieces"] // 4


def test_find_over_long_sequences(oracle):
    """kbo::find (lib.rs:808-821): the run lengths behind the characters, max_gap_len 0 and 50"""
    import torch
    rng = np.random.default_rng(79)
    contigs = _genome(rng, 300_000, contigs=2)
    sbwt, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([c.tobytes() for c in contigs], k=31)
    seqs = _sequences(rng, contigs, 30, [500, 3000, 20_000], 0.01, 0.002, 0.3)
    concat, offsets = _batch_of(seqs)
    exp_chars = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads())
    for gap in (0, 50):
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"), want_ms=False)
        dev.run_find(max_gap_len=gap, runs_per_seq=4000)
        torch.cuda.synchronize()
        assert dev.fused
        recs, first = dev.run_lengths_host()
        want_recs, want_first = oracle.run_lengths_batch(exp_chars, offsets, gap)
        assert np.array_equal(np.asarray(first, dtype=np.uint64), want_first), gap
        assert np.array_equal(np.asarray(recs, dtype=np.uint64).reshape(-1, 7), want_recs), gap
    # sequences of 1 and 2 bases in between (no alignment: derandomize.rs:274-276 asserts on fewer than 3 values - and no run, whatever the
    # caller's buffer held: here 'M's, which the generic run-length kernels would report as runs), at both lengths of the longest sequence
    # that decide which run-length kernel runs (<= 480: the LDS-staged one; beyond: one lane per sequence)
    for longest in (400, 20_000):
        part = [q for q in (_sequences(rng, contigs, 12, [200, 300, 380], 0.01, 0.0) if longest == 400 else seqs) if 161 <= len(q) <= longest][:12]
        assert len(part) >= 6
        mixed = []
        for i, q in enumerate(part):
            mixed.append(q)
            mixed.append(ACGT[rng.integers(0, 4, 1 + i % 2)])
        c2, o2 = _batch_of(mixed)
        exp2 = ora.matches_batch(np.concatenate(part), np.concatenate([[0], np.cumsum([len(q) for q in part])]).astype(np.uint64), 1e-7, n_threads=threads())
        for gap in (0, 50):
            dev = batch.DeviceBatch(sbwt, c2, o2, device=torch.device("cuda:0"), want_ms=False)
            dev.chars.fill_(ord("M"))
            dev.run_find(max_gap_len=gap, runs_per_seq=4000)
            torch.cuda.synchronize()
            assert dev.fused
            recs, first = dev.run_lengths_host()
            want_recs, want_first = oracle.run_lengths_batch(exp2, np.concatenate([[0], np.cumsum([len(q) for q in part])]).astype(np.uint64), gap)
            # (every tiny sequence follows its long one: it has the first-run index of the next long sequence and no run of its own)
            got_first = np.asarray(first, dtype=np.uint64)
            assert np.array_equal(got_first[0::2][:len(part) + 1], want_first), (longest, gap)
            assert np.array_equal(got_first[1::2], want_first[1:]), (longest, gap)
            assert np.array_equal(np.asarray(recs, dtype=np.uint64).reshape(-1, 7), want_recs), (longest, gap)


def test_host_entry_points_over_long_sequences(oracle):
    """kbo_matches_batch / kbo_map_batch / kbo_find_batch and the single-sequence kbo::matches / map / find over host buffers: the
    slabs of a batch with sequences of more than 160 bases take the kernel for sequences of any length (host_batch.cpp)"""
    rng = np.random.default_rng(83)
    contigs = _genome(rng, 400_000, contigs=2)
    sbwt, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=31, num_threads=threads()))
    sbwt.to_device(-1)  # (the plan structures now, not once the copy has seen the bases that pay for them)
    ora = oracle.Index.build([c.tobytes() for c in contigs], k=31)
    seqs = [q for q in _sequences(rng, contigs, 80, [5, 150, 161, 500, 3000, 20_000, 100_000], 0.01, 0.002, 0.3) if len(q) >= 3]
    concat, offsets = _batch_of(seqs)
    exp = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads())
    assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp)
    assert np.array_equal(batch.map_batch(sbwt, concat, offsets), np.frombuffer(oracle.relative_to_ref(concat, exp), dtype=np.uint8))
    for gap in (0, 50):
        rles, ro = batch.find_batch(sbwt, concat, offsets, kbo_amd.FindOpts(max_gap_len=gap))
        want_recs, want_first = oracle.run_lengths_batch(exp, offsets, gap)
        assert np.array_equal(ro, want_first), gap
        got = np.array([tuple(r) for r in rles], dtype=np.uint64).reshape(-1, 7) if isinstance(rles, list) else rles
        assert np.array_equal(got, want_recs), gap
    q = seqs[-1]
    assert kbo_amd.matches(q.tobytes(), sbwt) == [chr(v) for v in exp[int(offsets[-2]):]]


def test_map_stream_batches_in_flight(oracle):
    """kbo_map_stream_*: the library's own pipelines - batches of reads and of long sequences in turn, several rounds through the same
    slots, every base of every batch"""
    import torch
    rng = np.random.default_rng(85)
    contigs = _genome(rng, 400_000, contigs=2)
    sbwt, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=31, num_threads=threads()))
    sbwt.to_device(-1)
    ora = oracle.Index.build([c.tobytes() for c in contigs], k=31)
    dev0 = torch.device("cuda:0")
    sets = []
    for lengths, n, fmt, want_ms in (([150], 4000, True, False), ([700, 3000, 20_000], 40, False, False), ([60, 100, 159], 3000, False, True),
                                     ([10_000], 30, True, False)):
        seqs = [q for q in _sequences(rng, contigs, n, lengths, 0.01, 0.002, 0.0, spice=False) if len(q) >= 3]
        concat, offsets = _batch_of(seqs)
        exp, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads(), want_d=True)
        want = np.frombuffer(oracle.relative_to_ref(concat, exp), dtype=np.uint8) if fmt else exp
        sets.append((batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=fmt, want_ms=want_ms), want, exp_d))
    ms = batch.MapStream(sbwt, max(d.n_seqs for d, _, _ in sets), max(d.total for d, _, _ in sets), 0, pipelines=2)
    for rnd in range(3):
        tickets = []
        for d, _, _ in sets:
            d.chars.fill_(0xEE)
            d.ms.fill_(0xEE)
        torch.cuda.synchronize(dev0)
        for d, _, _ in sets:
            tickets.append(ms.submit(d))
        for (d, want, want_d), t in zip(sets, tickets):
            ms.wait(t)
            assert np.array_equal(d.chars[:d.total].cpu().numpy(), want), rnd
            if d.want_ms:  # (d_ms_out: the matching statistics of every base too, raw as kbo_ms_batch_dev gives them)
                assert np.array_equal(d.ms[:d.total].cpu().numpy(), want_d), rnd
    ms.sync()
    ms.close()


def test_map_stream_wait_for_a_ticket_whose_slot_was_taken_again(oracle):
    """kbo_map_stream_submit never blocks, so more batches than the pipelines have slots can be queued: waiting for an early ticket
    still means its batch is complete - its pipeline's later batches are behind it.  The first batch is held back by a stream that
    sleeps (ready_stream); the wait may not return before the sleep is over"""
    import torch
    rng = np.random.default_rng(86)
    contigs = _genome(rng, 200_000, contigs=1)
    sbwt, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=31, num_threads=threads()))
    sbwt.to_device(-1)
    ora = oracle.Index.build([c.tobytes() for c in contigs], k=31)
    dev0 = torch.device("cuda:0")
    seqs = _sequences(rng, contigs, 2000, [150], 0.01, 0.0, 0.0, spice=False)
    concat, offsets = _batch_of(seqs)
    want = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads())
    devs = [batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=False, want_ms=False) for _ in range(4)]
    ms = batch.MapStream(sbwt, devs[0].n_seqs, devs[0].total, 0, pipelines=2)
    for d in devs:  # (routes and lazy parts settled)
        ms.wait(ms.submit(d))
    for d in devs:
        d.chars.fill_(0xEE)
    torch.cuda.synchronize(dev0)
    held = torch.cuda.Stream(dev0)
    with torch.cuda.stream(held):
        torch.cuda._sleep(400_000_000)  # (about 0.2 s)
        woke = torch.cuda.Event()
        woke.record(held)
    tickets = [ms.submit(devs[i % 4], ready_stream=held if i == 0 else None) for i in range(13)]
    ms.wait(tickets[0])
    assert woke.query(), "kbo_map_stream_wait came back before the batch could have started"
    for v in (2, 3):  # (its pipeline's other batches whose slots were taken again)
        ms.wait(tickets[v])
    ms.sync()
    for d in devs:
        assert np.array_equal(d.chars[:d.total].cpu().numpy(), want)
    with pytest.raises(kbo_amd.KboError):
        ms.wait(tickets[-1] + 1)
    ms.close()


def test_the_walks_need_only_their_own_work_bytes(oracle):
    """kbo_ms_batch_dev over long sequences with a d_work of kbo_ms_work_bytes(): the regions of the kernels for long sequences that
    kbo_work_bytes() adds (about 0.5 B per base) are kbo_map_batch_dev's / kbo_find_batch_dev's - the walks never touch them and do not
    ask for them; kbo_map_batch_dev over the same batch still refuses the smaller buffer"""
    import ctypes as C
    import torch
    rng = np.random.default_rng(97)
    contigs = _genome(rng, 300_000, contigs=2)
    sbwt, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=31, num_threads=threads()))
    ora = oracle.Index.build([c.tobytes() for c in contigs], k=31)
    seqs = _sequences(rng, contigs, 20, [500, 3000, 20_000], 0.01, 0.002)
    concat, offsets = _batch_of(seqs)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"), want_ms=True)
    L = kbo_amd.lib()
    small = int(L.kbo_ms_work_bytes(dev.n_seqs, dev.total, dev.max_len, dev.k))
    assert small < dev.work_bytes and dev.work_bytes - small > dev.total // 4
    full = dev.work_bytes
    dev.work_bytes = small
    dev.walk()
    torch.cuda.synchronize()
    _, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads(), want_d=True)
    assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d)
    with pytest.raises(kbo_amd.KboError):
        dev.run()
    dev.work_bytes = full
    dev.run()
    torch.cuda.synchronize()
    assert not dev.fused  # (the MS values are wanted: the walk + the piece-wise derandomize / translate kernel, whose region the walk's figure leaves out)
    exp_chars = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads())
    assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars)
