"""BASELINE.json's two large configurations at the sizes where the code paths change, every read against the CPU oracle,
shipped defaults of every knob, plus a check of the index that passes through neither index builder:
   C4   kbo map, 250 Mbp index (2.5 * 10^8 rows: the 17-base depth table with anchors - margin 3.05 bases over log4(rows) -, 64 GiB
        grouped, 14-base seed table, redo pass over whole reads), the first 4 M of C4's reads (bench.py --config C4 makes
        the same ones) through kbo_map_batch (host slabs) and the device-resident path
   C5's code path   an index of 1.3 * 10^9 rows, k = 63 - what the 3 Gbp index of `kbo call` takes: no depth table (17 bases are
        less than log4(rows) + 1.9), contraction entries behind 64-bit offsets by SIZE (not forced), the 13-base seed table (>=
        512 Mi rows), units + the guided walk over recovery lines, call mode - 200 k reads of 150 bases and 2 000 of 10 kbp
        through kbo_ms_batch, kbo_call_walk_dev (every site) and kbo_call_batch (60 reads against oracle.call).  Needs 128 GB
        of host memory (skipped below that) and most of ten minutes, nearly all of it index building on the host.
The oracle adopts the product-built index at these sizes (its own row-sorting builder needs hours); what ties the index to the
text without either builder is gpu_helpers.check_rows_off_the_text (oracle/index_check.c), and error-free reads, whose depth is
min(i + 1, k) whatever any index says."""
import ctypes
import os
import time

import numpy as np
import pytest

import kbo_amd
from kbo_amd import batch, derandomize, synth
from gpu_helpers import adopt, call_walk_sites, check_rows_off_the_text, long_reads, oracle_sites, threads

pytestmark = pytest.mark.gpu


def _ramp_of_error_free_reads(sbwt, g, n_reads, k, seed):
    """reads copied off the text: depth min(i + 1, k) at base i, all 'M' (no oracle, no index in the expectation)"""
    concat, offsets = synth.reads(g, n_reads, 150, 0.0, seed=seed)
    d, _, _ = batch.ms_batch(sbwt, concat, offsets)
    ramp = np.minimum(np.arange(150) + 1, k).astype(np.uint8)
    assert np.array_equal(d.reshape(-1, 150), np.broadcast_to(ramp, (n_reads, 150)))
    assert (batch.matches_batch(sbwt, concat, offsets) == ord("M")).all()


def _bails(L, sbwt):
    b = ctypes.c_uint32(0)
    kbo_amd.check(L.kbo_index_plan_holdoff(sbwt._h, -1, b, None))
    return b.value


def test_c4_at_its_real_size(oracle):
    import torch
    L = kbo_amd.lib()
    t0 = time.time()
    g = synth.genome(250_000_000)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    n = sbwt.n_sets()
    assert n == 250_000_001
    t_build = time.time() - t0
    ora, lcs = adopt(oracle, sbwt, parts=True)
    sbwt.to_device(-1)
    lay = sbwt.device_layout()
    assert lay["dtab_order"] == 17 and lay["dtab_grouped"] == 1 and lay["anchor_bytes"] > 0  # what 2.5 * 10^8 rows get by size
    assert lay["seed_depth"] == 14 and lay["entries_64bit"] == 0 and lay["pair_bytes"] > 0
    check_rows_off_the_text(oracle, ora, lcs, sbwt, g)
    _ramp_of_error_free_reads(sbwt, g, 200_000, 31, seed=77)
    concat, offsets = synth.reads(g, 4_000_000, 150, 0.01)  # (bench.py --config C4: rank 0's first 4 M reads)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads(), want_d=True)
    exp_map = np.frombuffer(oracle.relative_to_ref(concat, exp_chars), dtype=np.uint8)
    assert np.array_equal(batch.map_batch(sbwt, concat, offsets, format=True), exp_map)  # kbo_map_batch, slabs of 32 MiB
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"), format=True)
    dev.ms.fill_(0xEE)
    dev.run()
    torch.cuda.synchronize()
    assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d)
    assert np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_map)
    assert _bails(L, sbwt) == 0  # the plan with the table really ran (no launch gave it up)
    print("C4 size: build %.0f s, copy %s, whole test %.0f s" % (t_build, {k: round(v, 2) for k, v in lay.items() if k.endswith("seconds")},
                                                                  time.time() - t0))


def _host_gb():
    try:
        return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 1e9
    except (ValueError, OSError):
        return 0.0


def test_c5_code_path(oracle):
    import torch
    if _host_gb() < 128:
        pytest.skip("needs 128 GB of host memory (index of 1.3 * 10^9 rows, k = 63)")
    L = kbo_amd.lib()
    k, G = 63, 1_300_000_000
    t0 = time.time()
    g = synth.genome(G, seed=0xC5C5)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=k, num_threads=threads()))
    n = sbwt.n_sets()
    assert n == G + 1
    t_build = time.time() - t0
    ora, lcs = adopt(oracle, sbwt, parts=True)
    sbwt.to_device(-1)
    lay = sbwt.device_layout()
    # by SIZE, nothing forced: no depth table, 64-bit entry offsets, the 13-base seed table, recovery lines, no two-base blocks
    assert lay["dtab_order"] == 0 and lay["entries_64bit"] == 1 and lay["seed_depth"] == 13 and lay["lines_bytes"] >= 2 * n
    assert lay["pair_bytes"] == 0
    t_copy = time.time() - t0
    check_rows_off_the_text(oracle, ora, lcs, sbwt, g, n_samples=6000)
    _ramp_of_error_free_reads(sbwt, g, 50_000, k, seed=78)
    thr = derandomize.random_match_threshold(k, sbwt.n_kmers(), 4, 1e-7)
    rng = np.random.default_rng(55)
    c1, o1 = synth.reads(g, 200_000, 150, 0.01, seed=0xC5C6)
    c2, o2 = long_reads(rng, g, 2000, 10_000, 0.01)
    for concat, offsets in ((c1, o1), (c2, o2)):
        exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=threads(), want_d=True)
        d, _, _ = batch.ms_batch(sbwt, concat, offsets)  # units + the guided walk over recovery lines, 64-bit entries
        assert np.array_equal(d, exp_d)
        assert np.array_equal(batch.matches_batch(sbwt, concat, offsets), exp_chars)
        want = oracle_sites(ora, concat, offsets, thr)
        assert len(want) > 1000
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
        for plan in (1, 0):  # call mode of the plan-guided walk, then of the plain walk
            L.kbo_set_plan(plan, 0, 0)
            sites, ms, ok = call_walk_sites(L, sbwt, dev, thr)
            assert ok and np.array_equal(ms, exp_d), plan
            assert sites == want, plan
        L.kbo_set_plan(1, 0, 0)
        del dev
    assert _bails(L, sbwt) == 0
    opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=k, build_select=True))
    got = batch.call_batch(sbwt, c2[:10_000 * 60], o2[:61], opts)
    n_var = 0
    for s in range(60):
        exp, _, _ = ora.call(c2[10_000 * s:10_000 * (s + 1)].tobytes(), k, 1e-7)
        assert [(v.query_pos, bytes(v.query_chars).decode(), bytes(v.ref_chars).decode()) for v in got[s]] == exp, s
        n_var += len(exp)
    assert n_var > 1000  # (k = 63 leaves room above the threshold: the sites do resolve into variants)
    print("C5 path: build %.0f s, copy after %.0f s %s, whole test %.0f s, threshold %d, %d variants in 60 reads" %
          (t_build, t_copy, {k_: round(v, 2) for k_, v in lay.items() if k_.endswith("seconds")}, time.time() - t0, thr, n_var))
