"""oracle/plan_model.c — the CPU model of the product's plan-guided A1 stage, checked on the CPU:
its MS values (produced the stage's way: seed, diagonal, predicted values, units walked until they converge, redo) must
equal the literal walk of the oracle on every base, whatever its parameters decide; its counts must add up.  The path
cover comes from the product's host-side builder (kbo_index_path_cover; tests/test_path_cover.py checks its claims).
tests/test_gpu_model.py pins the model's counts to the kernels' own counters on the GPU."""
import numpy as np
import pytest

import kbo_amd
from kbo_amd import synth


def _adopt(oracle, sbwt):
    rows, Carr, lcs = sbwt.export_parts()
    return oracle.Index.from_parts(sbwt.k(), sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)


def _reads(rng, cat, n_reads, rate):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for r in range(n_reads):
        L = int(rng.choice([3, 31, 64, 100, 150, 151, 250]))
        a = int(rng.integers(0, len(cat) - L))
        p = cat[a:a + L].copy()
        hit = rng.random(L) < rate
        p[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
        if r % 13 == 0:
            p[int(rng.integers(0, L))] = ord("N")
        if r % 17 == 0 and L > 60:  # chimera
            b = int(rng.integers(0, len(cat) - L))
            p[L // 2:] = cat[b + L // 2:b + L]
        out.append(p)
    concat = np.concatenate(out)
    offsets = np.concatenate([[0], np.cumsum([len(p) for p in out])]).astype(np.uint64)
    return concat, offsets


@pytest.mark.parametrize("k", [5, 31, 64])
def test_model_ms_equals_the_literal_walk(oracle, k):
    rng = np.random.default_rng(40 + k)
    g = synth.genome(60_000, seed=500 + k)
    rep = np.tile(g[:400], 6)
    seqs = [np.concatenate([g, rep]).tobytes(), g[2000:9000].tobytes() + b"NN" + g[100:1500].tobytes()]
    sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, num_threads=2))
    ora = _adopt(oracle, sbwt)
    cover = sbwt.path_cover()
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    for rate in (0.0, 0.01, 0.05, 0.3):
        concat, offsets = _reads(rng, cat, 1500, rate)
        _, exp = ora.matches_batch(concat, offsets, 1e-3, n_threads=4, want_d=True)
        for fat in (0, 1):
            for seed_tab, seed_depth, gap, chunk, bail in ((8 if k >= 8 else 0, 11, 17, 32, 50), (0, 3, 2, 16, 0xFFFF),
                                                          (min(k, 10), 14, 24, 64, 0xFFFF), (4, 1, 5, 32, 0xFFFF)):
                P = oracle.PlanParams(seed_table_depth=seed_tab, seed_depth=seed_depth, seed_cap=64, gap=gap, chunk=chunk,
                                      list_cap=13, bail_x16=bail, recovery_lines=fat, depth_table=0, depth_anchors=0)
                ms, cn = ora.plan_model(cover, P, concat, offsets, n_threads=3)
                assert np.array_equal(ms, exp), (k, rate, fat, seed_tab, seed_depth, gap, chunk, bail)
                assert cn["bases"] == len(concat) and cn["items"] == len(offsets) - 1
                if fat == 0:  # the depth-table form (no units): orders from "resolves next to nothing" to "k itself"
                    for order, anch in ((2, 1), (6, 0), (6, 1), (11, 0), (11, 1), (17, 1)):
                        P.depth_table, P.depth_anchors = order, anch
                        ms, ct = ora.plan_model(cover, P, concat, offsets, n_threads=3)
                        assert np.array_equal(ms, exp), (k, rate, order, seed_tab, seed_depth)
                        assert ct["units"] == 0 and ct["tab_written"] <= ct["tab_lookups"] + 32
                        assert ct["items_noplan"] == ct["items_unseeded"]
                        if min(order, k) == k:  # a table of k bases knows every value (but for the first bytes of the buffer and
                            n_reads_with_n = sum(1 for r in range(len(offsets) - 1)  # the reads with a byte that is no base)
                                                 if not set(concat[int(offsets[r]):int(offsets[r + 1])].tolist()) <= set(b"ACGT"))
                            assert ct["tab_flagged"] <= ct["items_list_overflow"] + n_reads_with_n + 1 and ct["tab_anchored"] == 0
                    P.depth_table = P.depth_anchors = 0
                if cn["gave_up"]:
                    assert cn["units"] == 0 and cn["redo_bases"] == len(concat)
                else:
                    assert cn["units"] == cn["units_counted"]
                    assert cn["walk_out_bytes"] <= cn["walk_accepted"]
                    assert cn["unit_distinct_lines"] >= 4 * cn["units"] - cn["units_head"]


def test_model_on_the_bench_shape(oracle):
    """C2's shape at a tenth of its size: nearly every read seeds cleanly, about 1.3 units per read, none flagged."""
    g = synth.genome(500_000)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=4))
    ora = _adopt(oracle, sbwt)
    concat, offsets = synth.reads(g, 20_000, 150, 0.01)
    _, exp = ora.matches_batch(concat, offsets, 1e-7, n_threads=4, want_d=True)
    P = oracle.shipped_plan_params(31, sbwt.n_sets())
    assert P.depth_table == 13 and P.depth_anchors == 1  # log4(500 k) = 9.5, + 3.2, rounded up; anchors: a small index (round 4)
    P.depth_anchors = 0
    ms, c0 = ora.plan_model(sbwt.path_cover(), P, concat, offsets, n_threads=4)
    assert np.array_equal(ms, exp) and not c0["gave_up"] and c0["tab_flagged"] < 0.04 * c0["items"]
    P.depth_anchors = 1
    ms, cn = ora.plan_model(sbwt.path_cover(), P, concat, offsets, n_threads=4)
    assert np.array_equal(ms, exp) and not cn["gave_up"] and cn["units"] == 0
    assert cn["tab_flagged"] < 0.01 * cn["items"] and cn["items_noplan"] < 0.01 * cn["items"]  # next to no read goes to the plain walk ...
    assert 0 < cn["tab_anchored"] < 0.2 * cn["items"]                     # (the bases deeper than the table knows are read off the text)
    assert 8 < cn["tab_lookups"] / cn["mismatches"] <= 14      # ... and a mismatch costs about log4(rows) + 2 look-ups
    assert cn["items_flagged"] == cn["tab_flagged"] and cn["redo_bases"] == 150 * cn["tab_flagged"]
    for fat in (0, 1):
        P = oracle.shipped_plan_params(31, sbwt.n_sets(), recovery_lines=fat, depth_table=0)
        assert (P.seed_table_depth, P.seed_depth, P.gap) == (8, 12, 18)
        ms, cn = ora.plan_model(sbwt.path_cover(), P, concat, offsets, n_threads=4)
        assert np.array_equal(ms, exp)
        assert not cn["gave_up"] and cn["items_unseeded"] < 50
        assert 1.1 < cn["units"] / cn["items"] < 1.6
        assert 8 < cn["walk_accepted"] / cn["units"] < 25
    # unrelated reads: nothing seeds, every read is walked as chunks; the plan is given up
    other = synth.genome(200_000, seed=99)
    concat, offsets = synth.reads(other, 5_000, 150, 0.0)
    _, exp = ora.matches_batch(concat, offsets, 1e-7, n_threads=4, want_d=True)
    ms, cn = ora.plan_model(sbwt.path_cover(), oracle.shipped_plan_params(31, sbwt.n_sets(), depth_table=0), concat, offsets, n_threads=4)
    assert np.array_equal(ms, exp) and cn["gave_up"] == 1
    # with the depth table every base of a read without a plan is looked up: reads that match nothing deeper than the table
    # knows are done without a walk
    ms, cn = ora.plan_model(sbwt.path_cover(), oracle.shipped_plan_params(31, sbwt.n_sets()), concat, offsets, n_threads=4)
    assert np.array_equal(ms, exp) and cn["gave_up"] == 0 and cn["items_noplan"] > 0.7 * cn["items"]  # (some seed by chance)
    assert cn["tab_lookups"] >= 150 * cn["items_noplan"] - 32 and cn["tab_flagged"] < 0.4 * cn["items"]
