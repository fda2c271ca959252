"""BASELINE.json's configurations as parity tests (every read against the CPU oracle, bit-exact), run with the SHIPPED
defaults of every knob unless a test says otherwise (tests/conftest.py resets them around every test):
   C1  10 kbp query vs 1 Mbp reference, kbo map with all defaults (fill_gaps + call_variants), product vs oracle.map
   C2  5 Mbp index, the full 1 M x 150 bp batch bench.py times (plan-guided and plain walk)
   C3  kbo find at its real size: 100 Mbp index (12-base seed table, seed depth 16, unit gap 22, recovery lines beyond the
       Infinity Cache, two-base blocks), 4 M reads through kbo_find_batch / kbo_map_batch / kbo_ms_batch, every read
   C3/C4 shape  26 Mbp index (>= 24 Mi rows: the real PAIR kernel and the recovery lines, not forced on), 1.2 M reads,
       MS / map / find, plain and plan-guided; call mode (kbo_call_walk_dev, every read; kbo_call_batch) on the same index
   BIG  the 64-bit-offset entry layout at 50 Mbp, MS / matches / intervals and call mode, every read
   seed tables of 9 .. 13 bases forced onto a small index (what 32 Mi / 512 Mi-row indexes get by size)
The oracle adopts the product-built index for the large ones (its own row-sorting builder needs minutes there; builder
equality is tests/test_builder_vs_oracle.py).  C4 at 250 Mbp x 100 M reads and C5 (3 Gbp) are not run here: DESIGN.md 8."""
import numpy as np
import pytest

import kbo_amd
from kbo_amd import batch, derandomize, synth
from gpu_helpers import adopt as _adopt, call_walk_sites, long_reads, oracle_sites, threads as _threads

pytestmark = pytest.mark.gpu


@pytest.fixture
def plan_restore():
    """(tests/conftest.py puts every knob back to the shipped defaults around every test)"""
    return kbo_amd.lib()


def _check_find_every_read(oracle, sbwt, concat, offsets, exp_chars, gap=0):
    """kbo_find_batch against format::run_lengths_gapped (oracle, literal) of the expected characters, every read"""
    rles, ro = batch.find_batch(sbwt, concat, offsets, kbo_amd.FindOpts(max_gap_len=gap))
    exp_r, exp_o = oracle.run_lengths_batch(exp_chars, offsets, gap)
    assert np.array_equal(np.asarray(ro, dtype=np.uint64), exp_o)
    assert np.array_equal(np.asarray(rles, dtype=np.uint64).reshape(-1, 7), exp_r)


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
    for plan in (1, 0):  # shipped defaults (seed depth log4(rows) + 3, 64-base seed search, automatic gap); then the plain walk
        L.kbo_set_plan(plan, 0, 0)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
        dev.ms.fill_(0xEE)
        dev.run()
        torch.cuda.synchronize()
        assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d), plan
        assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars), plan
        del dev
    assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars)  # host entry point, slabs


def test_c3_find_at_its_real_size(oracle, plan_restore):
    """BASELINE C3: kbo find, 100 Mbp index, 150 bp reads, 1 % substitutions, shipped defaults.  What only this size
    reaches: the 12-base seed table (>= 32 Mi rows), seed depth 16 / unit gap 22 (log4 of 10^8 rows), recovery lines of
    200 MB (beyond L2, around the Infinity Cache), two-base blocks of 267 MB.  4 M of C3's 10 M reads (same generator,
    same seed: the first 4 M), every read compared."""
    g = synth.genome(100_000_000)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    n = sbwt.n_sets()
    assert n == 100_000_001
    ora = _adopt(oracle, sbwt)
    concat, offsets = synth.reads(g, 4_000_000, 150, 0.01)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
    sbwt.to_device(-1)
    assert sbwt.device_pair_bytes() > 0
    assert sbwt.device_plan_bytes() - 11 * n >= (128 << 20)  # path cover 9 B/row + lines 2 B/row + the 12-base table
    _check_find_every_read(oracle, sbwt, concat, offsets, exp_chars)          # kbo_find_batch (slabs of 32 MiB)
    exp_map = np.frombuffer(oracle.relative_to_ref(concat, exp_chars), dtype=np.uint8)
    assert np.array_equal(batch.map_batch(sbwt, concat, offsets, format=True), exp_map)
    d, _, _ = batch.ms_batch(sbwt, concat, offsets)
    assert np.array_equal(d, exp_d)
    bails = __import__("ctypes").c_uint32(0)
    kbo_amd.check(plan_restore.kbo_index_plan_holdoff(sbwt._h, -1, bails, None))
    assert bails.value == 0  # the plan-guided walk really ran (no launch gave the plan up)
    plan_restore.kbo_set_plan(0, 0, 0)  # and the plain walk (PAIR kernel) on a part
    d, _, _ = batch.ms_batch(sbwt, concat[:150 * 500_000], offsets[:500_001])
    assert np.array_equal(d, exp_d[:150 * 500_000])


def _call_mode_every_read(L, oracle, ora, sbwt, g, seed):
    """kbo_call_walk_dev (plan-guided and plain call mode) against the oracle's first pass of call_variants on every read:
    150 bp reads and 10 kbp reads (chunked, with k warm-up and borrowed bases); then kbo_call_batch vs oracle.call."""
    import torch
    rng = np.random.default_rng(seed)
    k = sbwt.k()
    thr = derandomize.random_match_threshold(k, sbwt.n_kmers(), 4, 1e-7)
    c1, o1 = synth.reads(g, 300_000, 150, 0.01, seed=seed)
    c2, o2 = long_reads(rng, g, 1500, 10_000, 0.01)
    for concat, offsets in ((c1, o1), (c2, o2)):
        want = oracle_sites(ora, concat, offsets, thr)
        assert len(want) > 1000
        _, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
        for plan in (1, 0):
            L.kbo_set_plan(plan, 0, 0)
            sites, ms, ok = call_walk_sites(L, sbwt, dev, thr)
            assert ok
            assert np.array_equal(ms, exp_d), plan
            assert sites == want, plan
        del dev
    L.kbo_set_plan(1, 0, 0)
    opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=k, build_select=True))
    got = batch.call_batch(sbwt, c2[:10_000 * 60], o2[:61], opts)
    n_var = 0
    for s in range(60):
        exp, _, _ = ora.call(c2[10_000 * s:10_000 * (s + 1)].tobytes(), k, 1e-7)
        assert [(v.query_pos, bytes(v.query_chars).decode(), bytes(v.ref_chars).decode()) for v in got[s]] == exp, s
        n_var += len(exp)
    return n_var


def test_c3_c4_shape_pair_kernel_and_plan(oracle, plan_restore):
    """26 Mbp index (> 24 Mi rows: device copies carry two-base blocks, the plain walk is the PAIR kernel and the guided
    walk reads the recovery lines), 1.2 M reads: map with formatting and find, every read, plain and plan-guided, shipped
    defaults; then call mode on the same index."""
    L = plan_restore
    g = synth.genome(26_000_000, seed=4321)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    assert sbwt.n_sets() >= 24 << 20
    ora = _adopt(oracle, sbwt)
    concat, offsets = synth.reads(g, 1_200_000, 150, 0.01, seed=99)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
    exp_map = np.frombuffer(oracle.relative_to_ref(concat, exp_chars), dtype=np.uint8)
    sbwt.to_device(-1)
    assert sbwt.device_pair_bytes() > 0 and sbwt.device_plan_bytes() > 0
    for plan in (0, 1):
        L.kbo_set_plan(plan, 0, 0)
        d, _, _ = batch.ms_batch(sbwt, concat, offsets)
        assert np.array_equal(d, exp_d), plan
        assert np.array_equal(batch.map_batch(sbwt, concat, offsets, format=True), exp_map), plan
    _check_find_every_read(oracle, sbwt, concat, offsets, exp_chars)
    _check_find_every_read(oracle, sbwt, concat[:150 * 100_000], offsets[:100_001], exp_chars[:150 * 100_000], gap=5)
    # k = 31 with this many k-mers leaves little room between the threshold (24) and k: few sites resolve into variants,
    # the comparison is what counts
    _call_mode_every_read(L, oracle, ora, sbwt, g, seed=31)


def test_big_layout_at_50mbp(oracle, plan_restore):
    """The 64-bit-offset contraction-entry layout (what indexes beyond ~3 * 10^8 rows get by size) forced on at 50 Mbp:
    MS / matches / intervals, plain and plan-guided (shipped defaults), and call mode, every read."""
    L = plan_restore
    L.kbo_set_force_big_layout(1)
    g = synth.genome(50_000_000, seed=777)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    sbwt.to_device(-1)  # (with its plan structures: a copy made by a first query would wait for the bases that pay for them)
    assert sbwt.device_layout()["entries_64bit"] == 1 and sbwt.device_plan_bytes() > 0
    ora = _adopt(oracle, sbwt)
    concat, offsets = synth.reads(g, 300_000, 150, 0.02, seed=5)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
    for plan in (0, 1):
        L.kbo_set_plan(plan, 0, 0)
        d, _, _ = batch.ms_batch(sbwt, concat, offsets)
        assert np.array_equal(d, exp_d), plan
    assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars)
    d2, lo, hi = batch.ms_batch(sbwt, concat[:3000], offsets[:21], want_intervals=True)
    for s in range(20):
        od, olo, ohi = ora.matching_statistics(concat[150 * s:150 * s + 150].tobytes())
        assert np.array_equal(lo[150 * s:150 * s + 150], olo.astype(np.uint32)) and np.array_equal(hi[150 * s:150 * s + 150], ohi.astype(np.uint32))
    _call_mode_every_read(L, oracle, ora, sbwt, g, seed=32)


@pytest.mark.parametrize("depth", [9, 11, 12, 13])
def test_forced_seed_tables(oracle, plan_restore, depth):
    """plan_kernel's seed table at the depths that otherwise only indexes of >= 32 Mi (12 bases, 128 MiB) and >= 512 Mi rows
    (13 bases, 512 MiB) get, forced onto a 400 kbp index: seeds of D bases straddle two 16-byte query blocks whenever they
    start behind base 16 - D (every restart after a failed seed), reads with junk heads make them restart."""
    L = plan_restore
    L.kbo_set_seed_table_depth(depth)
    rng = np.random.default_rng(depth)
    g = synth.genome(400_000, seed=1300 + depth)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    sbwt.to_device(-1)
    assert sbwt.device_plan_bytes() - 11 * sbwt.n_sets() >= 8 * 4 ** depth  # the table is there, at that depth
    for sub, junk in ((0.01, 0), (0.04, 0), (0.01, 12), (0.0, 5)):
        concat, offsets = synth.reads(g, 30_000, 150, sub, seed=depth * 10 + junk)
        concat = concat.copy()
        if junk:  # the first bases of every third read replaced: the first seed fails, the next starts mid-block
            r = concat.reshape(-1, 150)
            r[::3, :junk] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (len(r[::3]), junk))]
        exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
        d, _, _ = batch.ms_batch(sbwt, concat, offsets)
        assert np.array_equal(d, exp_d), (depth, sub, junk)
        assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars), (depth, sub, junk)


def test_plan_structures_wait_for_the_bases_that_pay_for_them(oracle, plan_restore):
    """kbo's own call pattern is one map / find / call per index (lib.rs:553, 720-761): a copy made by a first query holds the
    rank blocks and the contraction entries only - 63 MB for a 5 Mbp index, not 7.6 GB of cover and tables - and takes the plain
    walk; kbo_index_to_device (or the bases that pay for them) adds the plan structures.  Same results either way."""
    L = plan_restore
    g = synth.genome(5_000_000)
    sbwt, lcs = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    ora = oracle.Index.build([g.tobytes()], k=31)
    ref = bytes(g[1_000_000:1_010_000].tobytes())
    got = kbo_amd.map(ref, sbwt, lcs, kbo_amd.MapOpts())              # one 10 kbp sequence on a fresh index
    assert got == ora.map(ref, 31, 1e-7, True, True, True)
    lay = sbwt.device_layout()
    assert sbwt.device_plan_bytes() == 0 and lay["cover_bytes"] == 0 and lay["dtab_bytes"] == 0
    assert lay["rank_bytes"] + lay["entry_bytes"] + lay["pair_bytes"] < 200 << 20
    concat, offsets = synth.reads(g, 100_000, 150, 0.01)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
    assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars)  # 15 Mbp: far from what 7.6 GB of tables cost
    assert sbwt.device_plan_bytes() == 0
    L.kbo_set_plan_lazy(20_000_000)                                     # a copy that has seen 20 Mbp makes them
    assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars)
    assert sbwt.device_plan_bytes() > 0 and sbwt.depth_table_order() == 15
    assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars)  # (now through the one kernel)
    d, _, _ = batch.ms_batch(sbwt, concat, offsets)
    assert np.array_equal(d, exp_d)
    # a budget the grouped table does not fit: the plain layout; one that nothing fits: no table, the host-built seed table
    for budget, want_order, want_grouped in ((6 << 30, 15, 0), (64 << 20, 0, 0)):
        L.kbo_set_plan_table_budget(budget)
        s2, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
        s2.to_device(-1)
        lay = s2.device_layout()
        assert (lay["dtab_order"], lay["dtab_grouped"]) == (want_order, want_grouped) and lay["seed_depth"] in (14, 10)
        assert lay["dtab_bytes"] + lay["seed_bytes"] <= max(budget, 16 << 20)
        assert np.array_equal(batch.matches_batch(s2, concat, offsets), exp_chars)
        del s2


def test_hold_off_is_per_index(oracle, plan_restore):
    """A batch that gives the plan up holds planning off for the next launches over THAT copy of THAT index only: an
    unrelated index on the same device keeps planning (round 2 kept one process-wide counter)."""
    import ctypes
    L = plan_restore
    g1, g2 = synth.genome(300_000, seed=41), synth.genome(300_000, seed=42)
    a, _ = kbo_amd.build([g1], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    b, _ = kbo_amd.build([g2], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    oa, ob = oracle.Index.build([g1.tobytes()], k=31), oracle.Index.build([g2.tobytes()], k=31)
    bad, bad_off = synth.reads(g1, 20_000, 150, 0.12, seed=1)    # 12 % substitutions: far above the bail-out
    good, good_off = synth.reads(g2, 20_000, 150, 0.01, seed=2)
    _, exp_bad = oa.matches_batch(bad, bad_off, 1e-7, n_threads=_threads(), want_d=True)
    _, exp_good = ob.matches_batch(good, good_off, 1e-7, n_threads=_threads(), want_d=True)

    def state(ix):
        bails, hold = ctypes.c_uint32(0), ctypes.c_int(0)
        kbo_amd.check(L.kbo_index_plan_holdoff(ix._h, -1, bails, hold))
        return bails.value, hold.value

    for _ in range(3):
        d, _, _ = batch.ms_batch(a, bad, bad_off)
        assert np.array_equal(d, exp_bad)
        d, _, _ = batch.ms_batch(b, good, good_off)
        assert np.array_equal(d, exp_good)
    bails_a, hold_a = state(a)
    bails_b, hold_b = state(b)
    assert bails_a >= 1 and hold_a > 0  # index a bailed and is held off ...
    assert bails_b == 0 and hold_b == 0  # ... index b never noticed
    L.kbo_set_plan(1, 0, 0)              # an explicit enable clears every hold-off
    d, _, _ = batch.ms_batch(a, good[:150 * 100], good_off[:101])
    assert state(a)[1] == 0


def test_call_batch_does_not_depend_on_the_slab_size(oracle, plan_restore):
    """35 Mbp of 10 kbp reads in one slab (64 MiB, set on the handle) and in slabs of 16 MiB: the same variants, equal to the
    oracle's literal kbo::call on a sample.  (One slab of that size used to get chunks of 267 bases - total >> 17 - and the call
    mode of the plan-guided walk is only exact with chunks of a multiple of four bases: walk_chunk now rounds.)"""
    rng = np.random.default_rng(17)
    k = 51
    g = synth.genome(4_000_000, seed=61)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=k, num_threads=_threads()))
    ora = oracle.Index.build([g.tobytes()], k=k)
    R, L = 3500, 10_000
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = np.stack([g[a:a + L] for a in rng.integers(0, len(g) - L - 8, R)])
    hit = rng.random((R, L)) < 0.01
    reads[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
    for r in range(0, R, 3):
        p = int(rng.integers(200, L - 200))
        reads[r, p:L - 3] = reads[r, p + 3:].copy()
    concat, offsets = reads.reshape(-1), np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=k, build_select=True))
    res16 = batch.call_batch_arrays(sbwt, concat, offsets, opts)
    sbwt.set_opts(slab_bytes=64 << 20)
    res64 = batch.call_batch_arrays(sbwt, concat, offsets, opts)
    assert int(res16["var_offsets"][-1]) > 50 * R
    for s in range(R):
        assert list(batch.variants_of(res16, s)) == list(batch.variants_of(res64, s)), s
    for s in rng.integers(0, R, 25):
        exp, _, _ = ora.call(reads[s].tobytes(), k, 1e-7)
        assert [(p, q.decode(), r.decode()) for p, q, r in batch.variants_of(res64, int(s))] == exp


def test_options_are_per_handle(oracle, plan_restore):
    """kbo_index_set_opts: two indexes of one process with different settings of what used to be process-wide only - plan on / off,
    the depth table's order, the slab size and the devices of host batches.  Same results from every one of them."""
    import torch
    g = synth.genome(300_000, seed=51)
    ora = oracle.Index.build([g.tobytes()], k=31)
    concat, offsets = synth.reads(g, 30_000, 150, 0.01, seed=3)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=_threads(), want_d=True)
    a, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    b, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    c, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=_threads()))
    a.set_opts(plan=0)
    b.set_opts(depth_table=11, depth_table_anchors=0, slab_bytes=1 << 20, devices=[0])
    assert a.get_opts()["plan"] == 0 and c.get_opts()["plan"] == kbo_amd._capi.OPT_INHERIT and b.get_opts()["devices"] == [0]
    for ix in (a, b, c):
        ix.to_device(-1)
    assert a.device_plan_bytes() == 0 and a.depth_table_order() == 0          # no cover, no tables: the plain walk
    assert b.depth_table_order() == 11 and b.device_layout()["anchor_bytes"] == 0
    assert c.depth_table_order() not in (0, 11) and c.device_layout()["anchor_bytes"] > 0  # the shipped choice for this size (small: anchors)
    fused = {}
    for name, ix in (("a", a), ("b", b), ("c", c)):
        assert np.array_equal(batch.matches_batch(ix, concat, offsets), exp_chars), name  # (b: 5 slabs of 1 MiB)
        dev = batch.DeviceBatch(ix, concat, offsets, device=torch.device("cuda:0"), format=False, want_ms=True)
        dev.run()
        torch.cuda.synchronize()
        fused[name] = dev.fused
        assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars) and np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d), name
    assert fused == {"a": False, "b": True, "c": True}
    # switched off on a handle whose copy has the structures: its launches stop planning at once; switched on again: they plan again
    c.set_opts(plan=0)
    dev = batch.DeviceBatch(c, concat, offsets, device=torch.device("cuda:0"), format=False, want_ms=True)
    dev.run()
    torch.cuda.synchronize()
    assert not dev.fused and np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars)
    c.set_opts(plan=kbo_amd._capi.OPT_INHERIT)
    dev.run()
    torch.cuda.synchronize()
    assert dev.fused and np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars)
    # the process-wide switch still rules the handles that inherit, and only those
    plan_restore.kbo_set_plan(0, 0, 0)
    b.set_opts(plan=1)
    for ix, want in ((b, True), (c, False)):
        dev = batch.DeviceBatch(ix, concat, offsets, device=torch.device("cuda:0"), format=False, want_ms=True)
        dev.run()
        torch.cuda.synchronize()
        assert dev.fused == want and np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars)


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
        if r % 7 == 0:  # (exact copies stay exact: no variants expected of them below)
            reads.append(bytes(s))
            continue
        if r % 5 == 0:  # non-ACGT bytes: they split the sequence's own index into runs (runs shorter than k have no rows)
            for p in rng.integers(0, len(s), 3):
                s[p] = ord("N")
        if r % 11 == 0 and len(s) > 400:
            for p in range(100, 400, 37):  # a stretch of runs shorter than k
                s[p] = ord("N")
        if r % 13 == 0 and len(s) > 900:   # a copy of an earlier stretch of the same read (the automaton sees it twice)
            s[700:800] = s[100:200]
        if r % 3 == 0 and len(s) > 900:    # ... and a reverse-complemented one (what add_revcomp = true below is about)
            s[500:590] = bytes(reversed(bytes(s[250:340]).translate(bytes.maketrans(b"ACGTN", b"TGCAN"))))
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
    # the flat form (kbo_call_batch_flat: what the device's tail writes, plus - for the reads with N - the host's sites merged in by
    # query position) and rounds 3 - 5's route (every site to the host: kbo_set_call_device_emit(0)) give the same variants
    def as_lists(res):
        return [batch.variants_of(res, s) for s in range(len(reads))]
    want = [[(v.query_pos, bytes(v.query_chars), bytes(v.ref_chars)) for v in got[s]] for s in range(len(reads))]
    flat = batch.call_batch_arrays(sbwt, concat, offsets, opts)
    assert as_lists(flat) == want and int(flat["var_offsets"][-1]) == n_var
    try:
        kbo_amd.lib().kbo_set_call_device_emit(0)
        assert as_lists(batch.call_batch_arrays(sbwt, concat, offsets, opts)) == want
        kbo_amd.lib().kbo_set_call_device_emit(2)  # (the second pass's depths by the kernel for k > 64)
        assert as_lists(batch.call_batch_arrays(sbwt, concat, offsets, opts)) == want
    finally:
        kbo_amd.lib().kbo_set_call_device_emit(1)
    # ... and slabs cut everywhere (64 KiB: a few reads each, the two slots taking them in turn)
    sbwt.set_opts(slab_bytes=64 << 10)
    assert as_lists(batch.call_batch_arrays(sbwt, concat, offsets, opts)) == want
    sbwt.set_opts(slab_bytes=0)
    # the sequence's own index with its reverse complements (CallOpts.sbwt_build_opts.add_revcomp): the batch - the device's second
    # pass looks the reverse strand up in the same tables - against the single-sequence entry point, which builds that index
    opts_rc = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=k, build_select=True, add_revcomp=True))
    got_rc = batch.call_batch(sbwt, concat, offsets, opts_rc)
    differs = 0
    for s, rd in enumerate(reads):
        mine = [(v.query_pos, bytes(v.query_chars), bytes(v.ref_chars)) for v in got_rc[s]]
        one = kbo_amd.call(sbwt, lcs, rd, opts_rc)
        assert mine == [(v.query_pos, bytes(v.query_chars), bytes(v.ref_chars)) for v in one], s
        differs += mine != [(v.query_pos, bytes(v.query_chars), bytes(v.ref_chars)) for v in got[s]]
    assert differs > 0  # (the reverse-complemented stretches change some calls: the option is not a no-op here)


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
    # kbo_call_batch from ONE host thread on device 0, then on device 1: the thread's cached device buffers (the arena its
    # per-sequence indexes borrow, the small batch's buffers) must not be reused across devices
    opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=31, build_select=True))
    rng = np.random.default_rng(8)
    c2, o2 = long_reads(rng, g, 40, 5000, 0.004)
    per_dev = []
    for dv in (0, 1, 0):
        with torch.cuda.device(dv):
            got = batch.call_batch(sbwt, c2, o2, opts)
            per_dev.append([[(v.query_pos, bytes(v.query_chars), bytes(v.ref_chars)) for v in vs] for vs in got])
    assert per_dev[0] == per_dev[1] == per_dev[2]
    for s in range(40):
        exp, _, _ = ora.call(c2[5000 * s:5000 * (s + 1)].tobytes(), 31, 1e-7)
        assert [(q, a.decode(), b.decode()) for q, a, b in per_dev[1][s]] == exp


def test_device_made_layout_equals_the_hosts(plan_restore):
    """a device copy's rank blocks, contraction entries and two-base blocks are made on the device from the row bit-vectors and the LCS bytes
    (layout_kernels.hip), its path cover from those (cover_kernels.hip): byte for byte the host's make_device_layout / make_path_cover - 32-bit entries in the arena, 64-bit entries by force, two-base
    blocks, an index with repeats (long stretches of equal LCS values) and one of a single short sequence"""
    import ctypes as C
    import torch
    L = kbo_amd.lib()
    rng = np.random.default_rng(7)
    g = synth.genome(700_000, seed=99)
    rep = np.concatenate([g[:200_000], np.tile(g[1000:1400], 50), g[200_000:300_000], np.full(300, ord("A"), dtype=np.uint8), g[300_000:400_000]])
    cases = [("plain", [g], 31, 0, 0), ("pairs", [g], 31, 1, 0), ("big", [g[:300_000]], 51, 0, 1), ("repeats", [rep, g[:5000]], 21, 1, 0),
             ("tiny", [g[:40]], 11, 0, 0), ("many short", [g[i:i + 37] for i in range(0, 20_000, 50)], 15, 1, 1),
             ("a cycle", [np.tile(g[5000:5050], 40), g[:3000]], 13, 0, 0), ("contigs", [g[i:i + 20_000] for i in range(0, 600_000, 23_000)], 31, 0, 0)]
    for name, seqs, k, pairs, big in cases:  # (tests/conftest.py puts the knobs back behind the test)
        L.kbo_set_pair_steps(0 if pairs else (1 << 63), 4)
        L.kbo_set_force_big_layout(big)
        sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, num_threads=_threads()))
        with torch.cuda.device(0):
            sbwt.to_device(-1)
            diff = C.c_uint64(123)
            kbo_amd.check(L.kbo_index_layout_check(sbwt._h, -1, C.byref(diff)))
            cdiff = C.c_uint64(123)  # ... and the path cover the copy laid out on the device (cover_kernels.hip), position for position
            kbo_amd.check(L.kbo_index_cover_check(sbwt._h, C.byref(cdiff)))
        assert diff.value == 0, (name, diff.value)
        assert cdiff.value == 0, (name, cdiff.value)
        lay = sbwt.device_layout()
        assert bool(lay["pair_bytes"]) == bool(pairs and not big), name
