#!/usr/bin/env python3
"""Randomised run of the refinement stages (call, fill_gaps, full map) against the independent C oracle
on variant-laden reference/query pairs of random size, variant spacing, variant length and k.  Where the
reference would panic both sides must refuse.  SECONDS= wall budget, SEED= first seed."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbo_amd  # noqa: E402
from oracle import binding as ora  # noqa: E402

budget = float(os.environ.get("SECONDS", 120))
seed0 = int(os.environ.get("SEED", 1))
t_end = time.time() + budget
ACGT = b"ACGT"


def variant_pair(rng, n, spacing, max_len):
    ref = rng.choice(list(ACGT), size=n).astype(np.uint8).tobytes()
    q = bytearray()
    i = 0
    while i < len(ref):
        if 150 < i < len(ref) - 150 and i % spacing == 0:
            kind = int(rng.integers(0, 4))
            if kind == 0:
                q.append(ACGT[(ACGT.index(ref[i]) + 1 + int(rng.integers(0, 3))) % 4]); i += 1
            elif kind == 1:
                q += rng.choice(list(ACGT), size=int(rng.integers(1, max_len + 1))).astype(np.uint8).tobytes()
            elif kind == 2:
                i += int(rng.integers(1, max_len + 1))
            else:  # a longer stretch missing from the query: a gap for fill_gaps
                i += int(rng.integers(20, 200))
        else:
            q.append(ref[i]); i += 1
    return ref, bytes(q)


def both(f_prod, f_ora, what):
    try:
        e = f_ora()
        e_err = None
    except Exception as ex:  # noqa: BLE001
        e, e_err = None, ex
    try:
        g = f_prod()
        g_err = None
    except kbo_amd.KboError as ex:
        g, g_err = None, ex
    if (e_err is None) != (g_err is None):
        raise SystemExit(f"{what}: oracle {'refused' if e_err else 'accepted'} ({e_err}), product {'refused' if g_err else 'accepted'} ({g_err})")
    return e, g


it = 0
n_refused = 0
while time.time() < t_end:
    it += 1
    rng = np.random.default_rng(seed0 * 7919 + it)
    k = int(rng.choice([11, 15, 20, 25, 31, 41, 51]))
    n = int(rng.choice([800, 3000, 12000]))
    ref, q = variant_pair(rng, n, int(rng.choice([60, 120, 400])), int(rng.choice([1, 3, 6])))
    opts = kbo_amd.BuildOpts(k=k, build_select=True)
    sbwt, lcs = kbo_amd.build([q], opts)
    oi = ora.Index.build([q], k=k)
    p = float(rng.choice([1e-7, 1e-3, 1e-2]))
    tag = f"it {it}: k={k} n={n} p={p}"
    e, g = both(lambda: kbo_amd.call(sbwt, lcs, ref, kbo_amd.CallOpts(p, opts)), lambda: oi.call(ref, k, p)[0], "call " + tag)
    if e is None:
        n_refused += 1
    else:
        assert [(v.query_pos, bytes(v.query_chars).decode(), bytes(v.ref_chars).decode()) for v in g] == e, "call " + tag
    # the same call through kbo_call_batch (site windows from the device, run automaton instead of a per-sequence index)
    from kbo_amd import batch
    arr = np.frombuffer(ref, dtype=np.uint8)
    e2, g2 = both(lambda: batch.call_batch(sbwt, arr, np.array([0, len(ref)], dtype=np.uint64), kbo_amd.CallOpts(p, opts))[0],
                  lambda: oi.call(ref, k, p)[0], "call_batch " + tag)
    if e2 is not None:
        assert [(v.query_pos, bytes(v.query_chars).decode(), bytes(v.ref_chars).decode()) for v in g2] == e2, "call_batch " + tag
    for fg, cv, fmt in ((True, True, True), (True, False, False), (False, True, False)):
        mo = kbo_amd.MapOpts(max_error_prob=p, fill_gaps=fg, call_variants=cv, format=fmt, sbwt_build_opts=opts)
        e, g = both(lambda: kbo_amd.map(ref, sbwt, lcs, mo), lambda: oi.map(ref, k, p, fg, cv, fmt), f"map {fg}{cv}{fmt} " + tag)
        if e is not None:
            assert g == e, f"map fill_gaps={fg} call_variants={cv} format={fmt} " + tag
    if it % 20 == 0:
        print("ok", tag, flush=True)
print(f"{it} iterations ({n_refused} where the reference would panic and both sides refuse), all equal to the oracle")
