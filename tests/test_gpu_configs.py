"""BASELINE.json's configurations as parity tests (every read against the CPU oracle, bit-exact):
   C1  10 kbp query vs 1 Mbp reference, kbo map with all defaults (fill_gaps + call_variants), product vs oracle.map
   C2  5 Mbp index, the full 1 M x 150 bp batch bench.py times (plan-guided and plain walk)
   C3/C4 shape  an index above the two-base-step threshold (>= 24 Mi rows: the real PAIR kernel, not forced on),
       >= 1 M reads through kbo_map_batch / kbo_find_batch; the plan-guided walk on the same index
   BIG  the 64-bit-offset entry layout at 50 Mbp
The oracle adopts the product-built index for the large ones (its own row-sorting builder needs minutes there; builder
equality is tests/test_builder_vs_oracle.py).  C5 (3 Gbp) is not run: see DESIGN.md section 8."""
import numpy as np
import pytest

import kbo_amd
from kbo_amd import batch, synth

pytestmark = pytest.mark.gpu


def _adopt(oracle, sbwt):
    rows, Carr, lcs = sbwt.export_parts()
    return oracle.Index.from_parts(sbwt.k(), sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)


def _threads():
    import os
    return max(1, min(16, len(os.sched_getaffinity(0))))


@pytest.fixture
def plan_restore():
    L = kbo_amd.lib()
    yield L
    L.kbo_set_plan(1, 14, 40)
    L.kbo_set_force_big_layout(0)


def test_c1_map_full_defaults_10kbp_vs_1mbp(oracle):
    """lib.rs:720-761 with MapOpts::default(): the reference genome is the query of kbo::map's argument order
    (map(ref_seq, query_sbwt)); here a 10 kbp 'reference' stretch with variants against a 1 Mbp index."""
    rng = np.random.default_rng(11)
    g = synth.genome(1_000_000, seed=1234)
    sbwt, lcs = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    ref = bytearray(g[400_000:410_000].tobytes())
    for p in sorted(rng.integers(100, 9900, 25)):   # substitutions
        ref[p] = b"ACGT"[(b"ACGT".index(ref[p]) + 1 + int(rng.integers(0, 3))) % 4]
    del ref[5000:5007]                              # a deletion and an insertion
    ref[7000:7000] = b"GATTACAGATTACA"
    ref = bytes(ref)
    got = kbo_amd.map(ref, sbwt, lcs, kbo_amd.MapOpts())
    exp = ora.map(ref, 31, 1e-7, True, True, True)
    assert got == exp
    assert sum(1 for a, b in zip(got, ref) if a == b) > 9000  # (it is an alignment, not a row of gaps)


def test_c2_full_batch_every_read(oracle, plan_restore):
    import torch
    L = plan_restore
    g = synth.genome(5_000_000)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    concat, offsets = synth.reads(g, 1_000_000, 150, 0.01)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
    for plan in (1, 0):
        L.kbo_set_plan(plan, 14, 40)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
        dev.ms.fill_(0xEE)
        dev.run()
        torch.cuda.synchronize()
        assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d), plan
        assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars), plan
        del dev
    assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars)  # host entry point, slabs


def test_c3_c4_shape_pair_kernel_and_plan(oracle, plan_restore):
    """26 Mbp index (> 24 Mi rows: device copies carry two-base blocks and the plain walk is the PAIR kernel),
    1.2 M reads: map with formatting and find, every read, plain and plan-guided."""
    L = plan_restore
    g = synth.genome(26_000_000, seed=4321)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    assert sbwt.n_sets() >= 24 << 20
    ora = _adopt(oracle, sbwt)
    concat, offsets = synth.reads(g, 1_200_000, 150, 0.01, seed=99)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
    exp_map = np.frombuffer(oracle.relative_to_ref(concat, exp_chars), dtype=np.uint8)
    L.kbo_set_plan(1, 14, 40)
    sbwt.to_device(-1)
    assert sbwt.device_pair_bytes() > 0 and sbwt.device_plan_bytes() > 0
    for plan in (0, 1):
        L.kbo_set_plan(plan, 14, 40)
        d, _, _ = batch.ms_batch(sbwt, concat, offsets)
        assert np.array_equal(d, exp_d), plan
        assert np.array_equal(batch.map_batch(sbwt, concat, offsets, format=True), exp_map), plan
    rles, ro = batch.find_batch(sbwt, concat, offsets, kbo_amd.FindOpts(max_gap_len=0))
    rng = np.random.default_rng(3)
    for s in rng.integers(0, 1_200_000, 400):
        exp = oracle.run_lengths_gapped(exp_chars[offsets[s]:offsets[s + 1]].tobytes(), 0)
        assert [tuple(int(v) for v in r) for r in rles[ro[s]:ro[s + 1]]] == exp


def test_big_layout_at_50mbp(oracle, plan_restore):
    L = plan_restore
    L.kbo_set_force_big_layout(1)
    g = synth.genome(50_000_000, seed=777)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    ora = _adopt(oracle, sbwt)
    concat, offsets = synth.reads(g, 300_000, 150, 0.02, seed=5)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
    for plan in (0, 1):
        L.kbo_set_plan(plan, 14, 40)
        d, _, _ = batch.ms_batch(sbwt, concat, offsets)
        assert np.array_equal(d, exp_d), plan
    assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars)
    d2, lo, hi = batch.ms_batch(sbwt, concat[:3000], offsets[:21], want_intervals=True)
    for s in range(20):
        od, olo, ohi = ora.matching_statistics(concat[150 * s:150 * s + 150].tobytes())
        assert np.array_equal(lo[150 * s:150 * s + 150], olo.astype(np.uint32)) and np.array_equal(hi[150 * s:150 * s + 150], ohi.astype(np.uint32))


def test_call_batch_equals_per_sequence_call(oracle):
    """kbo_call_batch (first pass + breakpoint scan on the device for the whole batch) against kbo::call per sequence:
    the oracle's literal restatement, and the single-sequence entry point.  Reads with substitutions, insertions and
    deletions against a 300 kbp index; ragged lengths incl. sequences without a single variant."""
    rng = np.random.default_rng(2024)
    g = synth.genome(300_000, seed=606)
    k = 51  # (with k = 31 and this many k-mers the threshold leaves no room for a significant peak in front of a variant)
    sbwt, lcs = kbo_amd.build([g], kbo_amd.BuildOpts(k=k, num_threads=_threads()))
    ora = oracle.Index.build([g.tobytes()], k=k)
    reads = []
    for r in range(120):
        L = int(rng.choice([200, 1000, 2500, 6000]))
        a = int(rng.integers(0, len(g) - L - 50))
        s = bytearray(g[a:a + L].tobytes())
        if r % 7:  # (every 7th read is an exact copy: no variants)
            for p in sorted(rng.integers(40, L - 40, max(1, L // 300)), reverse=True):
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    s[p] = b"ACGT"[(b"ACGT".index(s[p]) + 1 + int(rng.integers(0, 3))) % 4]
                elif kind == 1:
                    del s[p:p + int(rng.integers(1, 6))]
                else:
                    s[p:p] = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 6))).astype(np.uint8))
        reads.append(bytes(s))
    concat = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=k, build_select=True))
    got = batch.call_batch(sbwt, concat, offsets, opts)
    assert len(got) == len(reads)
    n_var = 0
    for s, rd in enumerate(reads):
        exp, _, _ = ora.call(rd, k, 1e-7)
        mine = [(v.query_pos, bytes(v.query_chars).decode(), bytes(v.ref_chars).decode()) for v in got[s]]
        assert mine == exp, s
        if s % 10 == 0:
            one = kbo_amd.call(sbwt, lcs, rd, opts)
            assert mine == [(v.query_pos, bytes(v.query_chars).decode(), bytes(v.ref_chars).decode()) for v in one], s
        n_var += len(mine)
    assert n_var > 100
    assert all(len(got[s]) == 0 for s in range(0, len(reads), 7))


def test_host_batches_over_two_distinct_devices(oracle):
    """kbo_set_devices with two different GPUs: index replicated on both, slabs dealt round-robin, disjoint output
    slices, no exchange.  Skipped on boxes with one GPU (the round-end GPU box has one; the
    same path with the device list (0, 0) runs in test_host_batches_in_slabs)."""
    import ctypes
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    g = synth.genome(400_000, seed=52)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    concat, offsets = synth.reads(g, 60_000, 150, 0.01)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
    L = kbo_amd.lib()
    try:
        L.kbo_set_slab_bytes(1 << 20)
        devs = (ctypes.c_int * 2)(0, 1)
        kbo_amd.check(L.kbo_set_devices(devs, 2))
        assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars)
        d, _, _ = batch.ms_batch(sbwt, concat, offsets)
        assert np.array_equal(d, exp_d)
        rles, ro = batch.find_batch(sbwt, concat, offsets, kbo_amd.FindOpts(max_gap_len=0))
        for s in (0, 1, 30_000, 59_999):
            assert [tuple(int(v) for v in r) for r in rles[ro[s]:ro[s + 1]]] == \
                oracle.run_lengths_gapped(exp_chars[offsets[s]:offsets[s + 1]].tobytes(), 0)
    finally:
        L.kbo_set_devices(None, 0)
        L.kbo_set_slab_bytes(32 << 20)
