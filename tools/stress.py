#!/usr/bin/env python3
"""Randomised stress run (not part of the test suite): random k, index content, batch shapes, error rates,
slab sizes, two-base steps on/off and gap lengths; every result is compared with the oracle.
SECONDS= wall budget, SEED= first seed."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import kbo_amd  # noqa: E402
from kbo_amd import batch, synth  # noqa: E402
from oracle import binding as ora  # noqa: E402

budget = float(os.environ.get("SECONDS", 120))
seed0 = int(os.environ.get("SEED", 1))
t_end = time.time() + budget
L = kbo_amd.lib()
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
it = 0
while time.time() < t_end:
    it += 1
    rng = np.random.default_rng(seed0 * 100003 + it)
    k = int(rng.choice([3, 5, 11, 21, 31, 32, 47, 63, 64, 101, 255]))
    G = int(rng.choice([2_000, 30_000, 200_000]))
    g = synth.genome(G, seed=int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.3:  # repeats and a non-ACGT byte in the index
        g = np.concatenate([g, np.tile(g[:int(rng.integers(20, 500))], int(rng.integers(2, 20))), [ord("N")], g[::-1][:1000]]).astype(np.uint8)
    rc = bool(rng.random() < 0.2)
    pairs_on = bool(rng.random() < 0.5)
    L.kbo_set_pair_steps(0 if pairs_on else (1 << 63), int(rng.choice([1, 4, 16])))
    L.kbo_set_slab_bytes(int(rng.choice([1 << 16, 1 << 18, 32 << 20])))
    big = bool(rng.random() < 0.15)          # 64-bit-offset entry layout
    L.kbo_set_force_big_layout(int(big))
    # plan-guided walk: on for most runs, with its knobs thrown around (seed depth / cap, gap, chunk, bail-out, tiny
    # unit array); device copies made while it is off carry no path cover at all
    plan_on = bool(rng.random() < 0.8)
    L.kbo_set_plan(int(plan_on), int(rng.choice([-1, 1, 4, 10, 14, 20])), int(rng.choice([4, 16, 40, 64])))
    L.kbo_set_plan_tuning(int(rng.choice([-1, 2, 3, 8, 20, 40])), int(rng.choice([16, 32, 100])),
                          int(rng.choice([0, 8, 32, 50, 64, 0xFFFF])))
    L.kbo_set_plan_unit_cap_divisor(int(rng.choice([1, 1, 1, 4, 30])))
    L.kbo_set_guided_walk(int(rng.choice([0, 1, 8, 32])), int(rng.choice([-1, 0, 1])))
    # depth table: by index size / none / every order from "knows nothing" to 16 (32-lane groups), anchors on and off
    L.kbo_set_depth_table(int(rng.choice([0, 0, 0, -1, 1, 3, 6, 9, 12, 16])))
    L.kbo_set_depth_table_anchors(int(rng.choice([-1, 0, 1, 1])))
    two_workers = bool(rng.random() < 0.15)  # the batch spread over a device list (both entries GPU 0)
    import ctypes
    devs = (ctypes.c_int * 2)(0, 0)
    L.kbo_set_devices(devs if two_workers else None, 2 if two_workers else 0)
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=k, num_threads=4, add_revcomp=rc))
    oi = ora.Index.build([g.tobytes()], k=k, add_revcomp=rc)
    # batch: reads of one length, ragged reads, a few long pieces
    shape = rng.choice(["uniform", "ragged", "long", "mixed", "reads", "reads"])
    lens = []
    if shape in ("uniform", "mixed"):
        lens += [int(rng.choice([32, 100, 128, 150, 151, 250, 256, 480]))] * int(rng.integers(100, 3000))
    if shape in ("ragged", "mixed"):
        lens += list(rng.integers(3, 600, int(rng.integers(100, 3000))))
    if shape == "reads":  # what the one kernel takes: at most 160 bases
        lens += list(rng.integers(3, 158, int(rng.integers(100, 6000))))
    if shape in ("long", "mixed"):
        lens += list(rng.integers(481, 90_000, int(rng.integers(1, 12))))
    lens = [int(min(n, len(g) - 1)) for n in lens if n >= 3]
    rng.shuffle(lens)
    sub = float(rng.choice([0.0, 0.01, 0.05, 0.3]))
    indel = bool(rng.random() < 0.4)
    pieces = []
    for n in lens:
        s0 = int(rng.integers(0, len(g) - n))
        p = g[s0:s0 + n].copy()
        hit = rng.random(n) < sub
        p[hit] = ACGT[rng.integers(0, 4, int(hit.sum()))]
        if rng.random() < 0.05:
            p[rng.integers(0, n, max(1, n // 50))] = ord("N")
        if indel and n > 30 and rng.random() < 0.4:  # an insertion or a deletion of 1 - 3 bases, the length kept
            q = int(rng.integers(5, n - 5))
            w = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                p = np.concatenate([p[:q], p[q + w:], ACGT[rng.integers(0, 4, w)]])
            else:
                p = np.concatenate([p[:q], ACGT[rng.integers(0, 4, w)], p[q:n - w]])
        pieces.append(p)
    concat = np.concatenate(pieces)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    p_err = float(rng.choice([1e-7, 1e-3, 0.1]))
    try:
        exp_chars, exp_d = oi.matches_batch(concat, offsets, p_err, n_threads=8, want_d=True)
    except Exception as e:  # threshold <= 1 etc.: both sides must refuse
        try:
            batch.matches_batch(sbwt, concat, offsets, p_err)
            raise SystemExit(f"iteration {it}: oracle refused ({e}) but the product accepted")
        except kbo_amd.KboError:
            continue
    tag = f"it {it}: plan={plan_on} k={k} G={len(g)} rc={rc} pairs={pairs_on} big={big} workers={2 if two_workers else 1} shape={shape} n={len(lens)} sub={sub} p={p_err}"
    d, _, _ = batch.ms_batch(sbwt, concat, offsets)
    assert np.array_equal(d, exp_d), "MS " + tag
    got = batch.matches_batch(sbwt, concat, offsets, p_err)
    assert np.array_equal(got, exp_chars), "chars " + tag
    if rng.random() < 0.5:  # the packed entry points: 2-bit words in, 2-bit words / 28-byte run-length records out
        words, epos, ebyt = batch.pack_reads(concat, offsets)
        pout = batch.matches_batch_packed(sbwt, words, offsets, epos, ebyt, p_err)
        assert np.array_equal(batch.unpack_matches(pout, offsets), exp_chars), "packed chars " + tag
        pg = int(rng.choice([0, 2, 50]))
        prl, pro = batch.find_batch_packed(sbwt, words, offsets, epos, ebyt, kbo_amd.FindOpts(max_error_prob=p_err, max_gap_len=pg))
        er, eo = ora.run_lengths_batch(exp_chars, offsets, pg)
        assert np.array_equal(pro, eo) and np.array_equal(prl.reshape(-1, 7), er), "packed rle " + tag
    gap = int(rng.choice([0, 0, 3, 50]))
    rles, ro = batch.find_batch(sbwt, concat, offsets, kbo_amd.FindOpts(max_error_prob=p_err, max_gap_len=gap))
    for s in rng.integers(0, len(lens), 40):
        exp = ora.run_lengths_gapped(exp_chars[offsets[s]:offsets[s + 1]].tobytes(), gap)
        assert [tuple(int(v) for v in r) for r in rles[ro[s]:ro[s + 1]]] == exp, f"rle seq {s} gap {gap} " + tag
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"), max_error_prob=p_err,
                            format=bool(rng.random() < 0.5))
    if rng.random() < 0.3:
        dev.max_len = 0
        dev.work_bytes = int(L.kbo_work_bytes(dev.n_seqs, dev.total, 0, k))
        dev.work = torch.zeros(dev.work_bytes // 8 + 2, dtype=torch.int64, device="cuda:0")
    dev.run()
    torch.cuda.synchronize()
    want = np.frombuffer(ora.relative_to_ref(concat, exp_chars), dtype=np.uint8) if dev.format else exp_chars
    assert np.array_equal(dev.ms.cpu().numpy()[:len(concat)], exp_d), "dev MS " + tag
    assert np.array_equal(dev.chars.cpu().numpy()[:len(concat)], want), "dev chars " + tag
    if 0 < dev.max_len <= 160:  # the one kernel's direct form (no MS values), its second pass on a tail stream; the packed-native form
        tail = torch.cuda.Stream("cuda:0")
        d2 = batch.DeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"), max_error_prob=p_err, format=dev.format, want_ms=False)
        d2.chars.fill_(0xEE)
        d2.run(tail_stream=tail if rng.random() < 0.5 else None)
        torch.cuda.synchronize()
        assert np.array_equal(d2.chars.cpu().numpy()[:len(concat)], want), "direct form chars (fused %s) " % d2.fused + tag
        try:
            pb = batch.PackedDeviceBatch(sbwt, concat, offsets, device=torch.device("cuda:0"), max_error_prob=p_err)
            pb.run(tail_stream=tail if rng.random() < 0.5 else None)
            torch.cuda.synchronize()
            assert np.array_equal(pb.chars(), exp_chars), "packed-native chars " + tag
        except kbo_amd.KboError as e:
            assert e.code == -8, "packed-native: " + str(e) + " " + tag  # KBO_E_UNSUPPORTED: this copy cannot take that kernel
    if rng.random() < 0.15 and not rc:  # kbo::call over a few of the sequences as a batch vs the oracle, one by one
        pick = [int(x) for x in rng.integers(0, len(lens), 6) if lens[int(x)] >= 3]
        if pick:
            cc = np.concatenate([pieces[x] for x in pick])
            co = np.concatenate([[0], np.cumsum([lens[x] for x in pick])]).astype(np.uint64)
            opts = kbo_amd.CallOpts(max_error_prob=p_err, sbwt_build_opts=kbo_amd.BuildOpts(k=k, build_select=True))
            # product and oracle must agree on WHETHER the call fails (reference panics), not only on its result
            exp_all, ora_err = [], None
            try:
                for x in pick:
                    exp_all.append(oi.call(pieces[x].tobytes(), k, p_err)[0])
            except ora.OracleError as e:
                ora_err = e
            try:
                got_calls = batch.call_batch(sbwt, cc, co, opts)
            except kbo_amd.KboError as e:
                assert ora_err is not None, f"call: the product refused ({e}) what the oracle accepted " + tag
                got_calls = None
            if got_calls is not None:
                assert ora_err is None, f"call: the oracle refused ({ora_err}) what the product accepted " + tag
                for exp_calls, vs in zip(exp_all, got_calls):
                    assert [(v.query_pos, bytes(v.query_chars).decode(), bytes(v.ref_chars).decode()) for v in vs] == exp_calls, "call " + tag
    print("ok", tag, flush=True)
print(f"{it} iterations, all equal to the oracle")
