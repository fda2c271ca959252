"""The depth-table form of the plan-guided A1 (kbo_amd/csrc/dtab_kernels.hip): the table's entries against the oracle and
against a brute-force count of the input's substrings; MS values against the oracle with tables from "knows next to nothing"
(every read goes to the plain walk) to "knows everything" (order = k), both lane-group widths; the kernels' own counters
against the CPU model of the stage (oracle/plan_model.c)."""
import numpy as np
import pytest

import kbo_amd
from kbo_amd import batch, synth
from gpu_helpers import adopt, threads

pytestmark = pytest.mark.gpu


def _present(seqs, k, s):
    """bool[4^s]: the strings of s bases (2-bit digits, last base least significant) inside the ACGT-runs of >= k bases"""
    out = np.zeros(4 ** s, dtype=bool)
    code = np.full(256, 4, dtype=np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    for q in seqs:
        c = code[np.frombuffer(q, dtype=np.uint8)]
        bad = np.flatnonzero(c == 4)
        edges = np.concatenate([[-1], bad, [len(c)]])
        for a, b in zip(edges[:-1] + 1, edges[1:]):
            if b - a < k or b - a < s:
                continue
            run = c[a:b]
            key = np.zeros(len(run) - s + 1, dtype=np.int64)
            for t in range(s):
                key = key * 4 + run[t:len(run) - s + 1 + t]
            out[key] = True
    return out


@pytest.mark.parametrize("k,order", [(31, 8), (31, 0), (5, 9), (12, 12)])
def test_depth_table_entries(oracle, k, order):
    rng = np.random.default_rng(3 + order)
    g = synth.genome(30_000, seed=11 + k)
    seqs = [g[:20_000].tobytes(), g[15_000:].tobytes() + b"NN" + g[100:1500].tobytes(), np.tile(g[300:340], 5).tobytes(),
            b"ACGTACGTAC", bytes(rng.choice(list(b"AC"), 400).astype(np.uint8))]
    L = kbo_amd.lib()
    L.kbo_set_depth_table(order)
    sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, num_threads=2))
    tab, o = sbwt.depth_table()
    assert o == (min(order, k) if order else oracle.shipped_depth_table_order(k, sbwt.n_sets())) and len(tab) == 4 ** o
    # brute force: T = the largest s whose last-s-bases string is present; 0x80 | e for the strings that are present whole
    want = np.zeros(4 ** o, dtype=np.uint8)
    keys = np.arange(4 ** o, dtype=np.int64)
    for s in range(1, o + 1):
        p = _present(seqs, k, s)
        want[p[keys & (4 ** s - 1)]] = s if s < o else 0x80
    if o < k:
        p1 = _present(seqs, k, o + 1)
        for c in range(4):
            want[p1[c * 4 ** o + keys]] |= 1 << c
    assert np.array_equal(tab, want)
    for view in (1, 2):  # (the device table holds the entries of three consecutive bases in one line: every entry three times)
        assert np.array_equal(sbwt.depth_table(view=view)[0], want)
    # and the oracle's walk on the strings themselves (the reference's semantics, dummy rows and all)
    ora = oracle.Index.build(seqs, k=k)
    for key in np.concatenate([rng.integers(0, 4 ** o, 300), np.flatnonzero(tab & 0x80)[:300]]):
        s = bytes(b"ACGT"[(int(key) >> (2 * (o - 1 - t))) & 3] for t in range(o))
        d = int(ora.matching_statistics(s)[0][-1])
        e = int(tab[key])
        assert (o if e & 0x80 else e) == d, s
        if e & 0x80 and o < k:
            for c in range(4):
                assert int(ora.matching_statistics(b"ACGT"[c:c + 1] + s)[0][-1] == o + 1) == (e >> c) & 1
    L.kbo_set_depth_table(-1)
    assert sbwt.depth_table()[1] == 0  # (launches ignore the table while the knob is negative)


def _reads(rng, cat, n_reads):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for r in range(n_reads):
        n = int(rng.choice([3, 17, 40, 100, 150, 151, 250, 255, 301]))
        a = int(rng.integers(0, len(cat) - n))
        p = cat[a:a + n].copy()
        hit = rng.random(n) < [0.0, 0.01, 0.03, 0.15][r % 4]
        p[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
        if r % 11 == 0 and n > 40:
            p = np.concatenate([p[:20], p[23:30], np.frombuffer(b"GATTACA", dtype=np.uint8), p[30:]])
        if r % 13 == 0:
            p[int(rng.integers(0, len(p)))] = rng.choice(list(b"Nn$\x00"))
        if r % 17 == 0 and n > 60:  # chimera
            b = int(rng.integers(0, len(cat) - n))
            p[n // 2:] = cat[b:b + len(p) - n // 2]
        out.append(p)
    return out


@pytest.mark.parametrize("k", [3, 5, 31, 64])
def test_table_form_equals_the_oracle(oracle, k):
    """Every order from 1 to "k itself" gives the oracle's values: low orders flag nearly every read with a mismatch (the
    redo pass walks them), high ones resolve them all; order 16 / 17 use groups of 32 lanes.  Reads, ragged, with N's,
    indels and chimeras; long sequences (chunks of 830 bases with warm-up, unstaged plan kernel); the plain walk and the
    guided walk as references."""
    import torch
    rng = np.random.default_rng(60 + k)
    g = synth.genome(80_000, seed=500 + k)
    seqs = [np.concatenate([g, np.tile(g[:500], 8)]).tobytes(), g[2000:9000].tobytes() + b"NN" + g[100:1500].tobytes(),
            bytes(rng.choice(list(b"ACG"), 3000).astype(np.uint8))]
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    reads = _reads(rng, cat, 3000)
    long1 = cat[1000:41000].copy()
    long1[rng.integers(0, len(long1), 300)] = ord("A")
    reads += [long1, cat[40000:47000].copy()]
    concat = np.concatenate(reads)
    offsets = np.concatenate([[0], np.cumsum([len(p) for p in reads])]).astype(np.uint64)
    ora = oracle.Index.build(seqs, k=k)
    exp_chars, exp_d = ora.matches_batch(concat, offsets, 1e-7, n_threads=4, want_d=True)
    L = kbo_amd.lib()
    for order, anch in [(o, a) for o, a in ((1, 1), (4, 0), (4, 1), (9, 1), (0, -1), (13, 0), (13, 1), (16, 1), (17, 0)) if o <= max(k, 1)]:
        L.kbo_set_depth_table(order)
        L.kbo_set_depth_table_anchors(anch)  # (anchors: bases deeper than the table knows are read off the path-cover text)
        sbwt, _ = kbo_amd.build(seqs, kbo_amd.BuildOpts(k=k, num_threads=2))
        assert sbwt.to_device(-1).depth_table_order() == (min(order, k) if order else oracle.shipped_depth_table_order(k, sbwt.n_sets()))
        for _ in range(2):  # (a launch that gave the plan up holds the next one off: both must be exact)
            d, _, _ = batch.ms_batch(sbwt, concat, offsets)
            assert np.array_equal(d, exp_d), (k, order, anch)
        dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
        dev.ms.fill_(0xEE)
        dev.run()
        torch.cuda.synchronize()
        assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), exp_d) and np.array_equal(dev.chars[:dev.total].cpu().numpy(), exp_chars)
        # only the reads (staged plan kernel, no chunks), from offset 0 of the buffer (its first 31 bases have no window in front)
        n_r = 3000
        d, _, _ = batch.ms_batch(sbwt, concat[:int(offsets[n_r])], offsets[:n_r + 1])
        assert np.array_equal(d, exp_d[:int(offsets[n_r])]), (k, order)
        L.kbo_set_depth_table(-1)  # the same copy without its table: the guided walk
        d, _, _ = batch.ms_batch(sbwt, concat, offsets)
        assert np.array_equal(d, exp_d), (k, order, anch)


def _compare(oracle, sbwt, ora, concat, offsets, order, anchors):
    import torch
    L = kbo_amd.lib()
    L.kbo_set_plan_stats(1)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"))
    dev.ms.fill_(0xEE)
    dev.walk()
    torch.cuda.synchronize()
    st = dev.plan_stats()
    P = oracle.shipped_plan_params(sbwt.k(), sbwt.n_sets(), depth_table=order, depth_anchors=anchors)
    ms, cn = ora.plan_model(sbwt.path_cover(), P, concat, offsets, n_threads=threads())
    assert np.array_equal(dev.ms[:dev.total].cpu().numpy(), ms)
    want = {"seed_lookups": cn["seed_lookups"], "seed_extensions": cn["seed_extensions"], "mismatches": cn["mismatches"],
            "tab_lookups": cn["tab_lookups"], "tab_written": cn["tab_written"], "tab_flagged": cn["tab_flagged"],
            "tab_unresolved": cn["tab_flagged"], "tab_anchored": cn["tab_anchored"], "items_noplan": cn["items_noplan"],
            "gave_up": cn["gave_up"], "guard": 0,
            # (the redo list: a flagged read of 150 bases goes there as five pieces of at most 32 bases with their warm-up)
            "redo_entries": len(offsets) - 1 if cn["gave_up"] else 5 * cn["items_flagged"]}
    got = {k: st[k] for k in want}
    assert got == want, (order, got, want)
    return cn


@pytest.mark.parametrize("order,anchors", [(0, -1), (0, 1), (12, 1), (9, 0), (16, 1)])  # (12 bases: a third of the stretches go to the anchors)
def test_model_counts_equal_the_kernels_counters_table_form(oracle, order, anchors):
    g = synth.genome(2_000_000, seed=2024)
    kbo_amd.lib().kbo_set_depth_table(order)
    kbo_amd.lib().kbo_set_depth_table_anchors(anchors)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=31, num_threads=threads()))
    anchors = int(oracle.shipped_depth_table_anchors(31, sbwt.n_sets(), order or 14)) if anchors < 0 else anchors
    ora = adopt(oracle, sbwt)
    o = sbwt.to_device(-1).depth_table_order()
    assert o == (order or 14)  # log4(2 M) = 10.5, + 3.2, rounded up
    for sub, seed in ((0.01, 1), (0.03, 2), (0.0, 3)):
        concat, offsets = synth.reads(g, 60_000, 150, sub, seed=seed)
        kbo_amd.lib().kbo_set_plan(1, 0, 0)  # (a launch that gave the plan up - order 9 leaves most reads unresolved - holds the next ones off)
        cn = _compare(oracle, sbwt, ora, concat, offsets, o, anchors)
        if order == 0 and sub == 0.01:
            assert cn["tab_flagged"] < (0.01 if anchors else 0.05) * cn["items"] and 9 < cn["tab_lookups"] / cn["mismatches"] < 14
    other = synth.genome(300_000, seed=77)  # reads from elsewhere: nothing seeds, every base of every read is looked up
    concat, offsets = synth.reads(other, 20_000, 150, 0.0, seed=4)
    kbo_amd.lib().kbo_set_plan(1, 0, 0)
    cn = _compare(oracle, sbwt, ora, concat, offsets, o, anchors)
    assert cn["items_noplan"] > 0.7 * cn["items"] and cn["gave_up"] == (1 if o <= 12 else 0)  # (a table of 9 - 12 bases knows next to nothing about a 2 Mbp index)
