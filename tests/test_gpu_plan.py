"""The plan-guided walk (kbo_amd/csrc/plan_kernels.hip) against the plain walk and the CPU oracle: MS values must be
identical whatever the plan kernel decides.  Every knob that changes how much is walked is forced through its corner:
mismatch groups of one base (units that cannot vouch for their successors -> redo pass), tiny chunks, plans given up
(bail-out + hold-off), shallow seeds that put reads on wrong diagonals."""
import numpy as np
import pytest

import kbo_amd
from kbo_amd import batch, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def guided_walk_only():
    """these tests are about the units and the guided walk: no depth table (tests/test_gpu_dtab.py has that form)"""
    kbo_amd.lib().kbo_set_depth_table(-1)


def _mutate(rng, s, rate):
    b = bytearray(s)
    for i in range(len(b)):
        if rng.random() < rate:
            b[i] = b"ACGT"[rng.integers(0, 4)]
    return bytes(b)


@pytest.fixture
def plan_defaults():
    """(tests/conftest.py puts every knob back to the shipped defaults around every test)"""
    return kbo_amd.lib()


def _workload(rng, ref_seqs, n_reads):
    """ragged reads with substitutions / indels / junk / chimeras, short reads, and two long sequences (chunked)"""
    cat = np.frombuffer(b"".join(ref_seqs), dtype=np.uint8)
    reads = []
    for r in range(n_reads):
        L = int(rng.choice([3, 7, 31, 64, 100, 150, 151, 250, 301]))
        a = int(rng.integers(0, len(cat) - L))
        s = bytearray(_mutate(rng, cat[a:a + L].tobytes(), [0, 0.01, 0.03, 0.2][r % 4]))
        if r % 11 == 0 and L > 40:  # deletion, insertion
            del s[20:23]
            s[30:30] = b"GATTACA"
        if r % 13 == 0:
            s[int(rng.integers(0, len(s)))] = rng.choice(list(b"Nn$\x00"))
        if r % 17 == 0 and L > 60:  # chimera
            b = int(rng.integers(0, len(cat) - 40))
            s[L // 2:] = cat[b:b + len(s) - L // 2].tobytes()
        reads.append(bytes(s))
    long1 = bytearray(cat[1000:31000].tobytes())
    for p in rng.integers(0, len(long1), 150):
        long1[p] = b"ACGT"[rng.integers(0, 4)]
    reads.append(bytes(long1))
    reads.append(cat[40000:47000].tobytes())
    concat = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    return concat, offsets


@pytest.mark.parametrize("k", [3, 5, 31, 64, 160])  # (160: LCS bytes with the top bit set in the recovery lines)
def test_plan_guided_walk_equals_plain_walk_and_oracle(oracle, plan_defaults, k):
    L = plan_defaults
    rng = np.random.default_rng(100 + k)
    g = synth.genome(60_000, seed=300 + k)
    rep = np.tile(g[:500], 8)
    seqs = [np.concatenate([g, rep]).tobytes(), g[2000:9000].tobytes() + b"NN" + g[100:1500].tobytes(),
            bytes(rng.choice(list(b"ACG"), 3000).astype(np.uint8))]
    sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, num_threads=2))
    assert sbwt.n_sets() > 0
    ora = oracle.Index.build(seqs, k=k)
    concat, offsets = _workload(rng, seqs, 2500)
    _, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=4, want_d=True)
    L.kbo_set_plan(1, 14, 40)
    sbwt.to_device(-1)  # the device copy gets its path cover while the plan is enabled
    assert sbwt.device_plan_bytes() > 0
    L.kbo_set_plan(0, 14, 40)
    d_plain, _, _ = batch.ms_batch(sbwt, concat, offsets)
    assert np.array_equal(d_plain, exp_d)
    # (seed depth, seed cap, unit gap, chunk, bail-out, divisor of the unit array); -1 = the shipped automatic choice
    settings = [(-1, 64, -1, 32, 50, 1),  # the shipped defaults
                (14, 40, 24, 32, 0xFFFF, 1), (14, 40, 2, 16, 0xFFFF, 1), (1, 8, 5, 64, 0xFFFF, 1), (3, 48, 24, 32, 0xFFFF, 1),
                (14, 40, 24, 32, 0, 1), (14, 40, 24, 32, 0xFFFF, 40),  # plan given up; unit array 1/40 of its size: overflow path
                (14, 40, 24, 32, 0xFFFF, 1)]
    for fat in (0, 1):  # the guided walk over rank blocks + entries / over the recovery lines (a size-based choice otherwise)
        L.kbo_set_guided_walk(0, fat)
        for dmin, cap, gap, chunk, bail, div in settings:
            L.kbo_set_plan(1, dmin, cap)
            L.kbo_set_plan_tuning(gap, chunk, bail)
            L.kbo_set_plan_unit_cap_divisor(div)
            for _ in range(2):  # (the launch after a plan was given up is held off; the one after that plans again or not)
                d, _, _ = batch.ms_batch(sbwt, concat, offsets)
                assert np.array_equal(d, exp_d), (k, fat, dmin, cap, gap, chunk, bail, div)
    # intervals are only ever produced by the plain walk
    d2, lo, hi = batch.ms_batch(sbwt, concat[:600], np.array([0, 600], dtype=np.uint64), want_intervals=True)
    od, olo, ohi = ora.matching_statistics(concat[:600].tobytes())
    assert np.array_equal(d2, od.astype(np.uint8)) and np.array_equal(lo, olo.astype(np.uint32)) and np.array_equal(hi, ohi.astype(np.uint32))


def test_plan_guided_walk_device_resident_and_map(oracle, plan_defaults):
    """The device-resident entry points and the whole map path (A1 -> A5/A6) with the plan on."""
    import torch
    L = plan_defaults
    g = synth.genome(300_000, seed=77)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=4))
    ora = oracle.Index.build([g.tobytes()], k=31)
    for sub in (0.0, 0.01, 0.08):
        concat, offsets = synth.reads(g, 20_000, 150, sub, seed=int(sub * 1000) + 3)
        exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=4, want_d=True)
        for plan in (1, 0, 2):
            L.kbo_set_guided_walk(0, 1 if plan == 2 else 0)
            L.kbo_set_plan(plan, 14, 40)
            dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
            dev.ms.fill_(0xEE)
            dev.run()
            torch.cuda.synchronize()
            assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d), (sub, plan)
            assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars), (sub, plan)
            assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars), (sub, plan)


def test_plan_guided_walk_big_layout(oracle, plan_defaults):
    L = plan_defaults
    g = synth.genome(120_000, seed=78)
    try:
        L.kbo_set_force_big_layout(1)
        sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=2))
        ora = oracle.Index.build([g.tobytes()], k=31)
        concat, offsets = synth.reads(g, 5000, 150, 0.02)
        _, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=4, want_d=True)
        for fat in (0, 1):
            L.kbo_set_guided_walk(0, fat)
            d, _, _ = batch.ms_batch(sbwt, concat, offsets)
            assert np.array_equal(d, exp_d), fat
    finally:
        L.kbo_set_force_big_layout(0)


from gpu_helpers import call_walk_sites as _call_walk_sites  # noqa: E402


@pytest.mark.parametrize("k", [31, 51])
def test_call_mode_of_the_plan_guided_walk(oracle, plan_defaults, k):
    """The breakpoint scan of call_variants (variant_calling.rs:268-273) carried by the guided walk's units: same sites as
    the plain walk in call mode and as a host scan of the oracle's MS, on reads and on chunked long sequences."""
    import torch
    from kbo_amd import derandomize
    L = plan_defaults
    rng = np.random.default_rng(700 + k)
    g = synth.genome(400_000, seed=900 + k)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=k, num_threads=4))
    ora = oracle.Index.build([g.tobytes()], k=k)
    thr = derandomize.random_match_threshold(k, sbwt.n_kmers(), 4, 1e-7)
    cat = g
    seqs = []
    for r in range(600):  # reads of 150 .. 2000 bases with substitutions, some indels, a few junk stretches
        n = int(rng.choice([150, 300, 1000, 2000]))
        a = int(rng.integers(0, len(cat) - n))
        s = bytearray(_mutate(rng, cat[a:a + n].tobytes(), [0.003, 0.01, 0.03][r % 3]))
        if r % 7 == 0:
            del s[60:62]
        if r % 9 == 0:
            s[100:100] = b"ACGTTGCA"
        if r % 31 == 0:
            s[40:70] = bytes(rng.choice(list(b"ACGT"), 30).astype(np.uint8))
        seqs.append(bytes(s))
    seqs.append(_mutate(rng, cat[5000:65000].tobytes(), 0.01))  # chunked
    concat = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
    want = set()
    for si, s in enumerate(seqs[:200] + seqs[-1:]):
        base = int(offsets[si if si < 200 else len(seqs) - 1])
        d, lo, hi = ora.matching_statistics(s)
        for i in range(1, len(s)):
            if d[i] < d[i - 1] and d[i - 1] >= thr and d[i] < thr:
                for j in range(i + 1, min(i + k + 1, len(s))):
                    if d[j] >= thr and hi[j] - lo[j] == 1:
                        want.add((base + i, base + j, int(lo[j])))
                        break
    exp_d = np.concatenate([ora.matching_statistics(s)[0] for s in seqs]).astype(np.uint8)
    got = {}
    # plain walk in call mode; guided walk over rank blocks + entries; over the recovery lines; with a unit array of 1/40 of
    # its size (most items overflow it and go to the redo pass in pieces, the sites their units wrote are void)
    for plan in (0, 1, 2, 3):
        L.kbo_set_plan(1 if plan else 0, 14, 40)
        L.kbo_set_guided_walk(0, 1 if plan == 2 else 0)
        L.kbo_set_plan_tuning(20, 32, 0xFFFF)
        L.kbo_set_plan_unit_cap_divisor(40 if plan == 3 else 1)
        sites, ms, ok = _call_walk_sites(L, sbwt, dev, thr)
        assert ok
        assert np.array_equal(ms, exp_d), plan
        got[plan] = sites
        lo_b, hi_b = int(offsets[200]), int(offsets[len(seqs) - 1])
        assert {x for x in sites if x[0] < lo_b or x[0] >= hi_b} == want, plan
    assert got[0] == got[1] == got[2] == got[3]
    assert len(got[0]) > 500
