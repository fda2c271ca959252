"""bench.py --call / --config C5: kbo call's first pass (the MS walk whose lanes run the breakpoint scan) and kbo_call_batch."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .legs import *  # noqa: F401,F403


def main_call(args):
    """kbo call, first pass (variant_calling.rs:266-273) over a batch of long reads resident in HBM: the walk in call mode
    (its lanes run the breakpoint scan); what leaves the device is one 16-byte record per site.  Parity: the sites of
    every read against the oracle's first pass."""
    import torch
    import kbo_amd
    from kbo_amd import batch, derandomize, synth
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cores, _ = usable_cores()
    genome, sbwt = build_or_load_index(args, cores)
    concat, offsets = synth.reads(genome, args.reads, args.read_len, args.sub_rate)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=device)
    thr = derandomize.random_match_threshold(args.k, sbwt.n_kmers(), 4, 1e-7)
    LISTS = 256  # KBO_CALL_LISTS
    cap = (dev.total // 8 + 4096) // LISTS * LISTS
    sites = torch.zeros((cap, 4), dtype=torch.int32, device=device)
    count = torch.zeros(LISTS * 16 + 16, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device)
    L = kbo_amd.lib()

    def step():
        # the walk in call mode: MS values + sites in one launch (variant_calling.rs:266-273), no intervals written
        kbo_amd.check(L.kbo_call_walk_dev(sbwt._h, dev.q.data_ptr(), dev.off.data_ptr(), dev.n_seqs, dev.total, dev.max_len,
                                          thr, dev.ms.data_ptr(), sites.data_ptr(), cap, count.data_ptr(),
                                          dev.work.data_ptr(), dev.work_bytes, stream.cuda_stream))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record(stream)
        step()
        ev[s][1].record(stream)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    walk_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    counts_all = count.cpu().numpy()
    counts = counts_all[:LISTS * 16:16]
    n_sites, seg = int(counts.sum()), cap // LISTS
    fits = bool((counts <= seg).all()) and int(counts_all[LISTS * 16]) == 0
    sites_h = sites.cpu().numpy().view(np.uint32)
    raw = np.concatenate([sites_h[g * seg:g * seg + min(int(counts[g]), seg)] for g in range(LISTS)]) if n_sites else sites_h[:0]
    raw = raw[raw[:, 0] != 0xFFFFFFFF]  # (void records: kbo_hip.h, kbo_call_walk_dev)
    n_sites = len(raw)
    exact = None
    if not args.no_cpu_baseline:  # every read: the oracle's first pass of call_variants (ora_call_sites_batch)
        from oracle import binding as ora
        rows, Carr, lcs = sbwt.export_parts()
        oi = ora.Index.from_parts(args.k, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
        recs = oi.call_sites_batch(concat, offsets, thr, n_threads=cores)
        base = offsets[recs[:, 0].astype(np.int64)]
        want = np.stack([base + recs[:, 1], base + recs[:, 2], recs[:, 3]], axis=1).astype(np.uint64)
        got = raw[:, :3].astype(np.uint64)
        want = want[np.lexsort(want.T[::-1])]
        got = got[np.lexsort(got.T[::-1])]
        exact = bool(fits and want.shape == got.shape and np.array_equal(want, got))
    # the whole of kbo::call over the same reads through the product entry point (host sequences in, variants out): first pass, second
    # pass on the device (call_second_kernels.hip), the host slicing the variants' characters; a sample against the oracle's literal call
    whole = None
    try:
        opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=args.k, build_select=True))
        best, res = 1e9, None
        for _ in range(2):
            t1 = time.perf_counter()
            res = batch.call_batch_arrays(sbwt, concat, offsets, opts)
            best = min(best, time.perf_counter() - t1)
        whole = {"entry_point": "kbo_call_batch", "ms": round(best * 1e3, 2), "us_per_read": round(best / args.reads * 1e6, 2),
                 "mbp_per_s": round(dev.total / best / 1e6, 1), "variants": int(res["var_offsets"][-1]),
                 "note": "host sequences in, variants out (the Python wrapper's copies of the records included)"}
        if not args.no_cpu_baseline:
            rng = np.random.default_rng(1)
            pick = [int(x) for x in rng.integers(0, args.reads, min(40, args.reads))]
            ok = True
            for s_ in pick:
                a_, b_ = int(offsets[s_]), int(offsets[s_ + 1])
                exp_calls, _, _ = oi.call(concat[a_:b_].tobytes(), args.k, 1e-7)
                ok = ok and [(p_, q_.decode(), r_.decode()) for p_, q_, r_ in batch.variants_of(res, s_)] == exp_calls
            whole["equal_to_oracle_call_on_sampled_reads"] = len(pick) if ok else False
    except kbo_amd.KboError as e:  # (a threshold the reference refuses, a sharded index: said, not hidden)
        whole = {"error": str(e)}
    bases = dev.total
    # ---- roofline of the first pass: SURVEY.md 8(d)'s bytes of the reference algorithm - 64 B per distinct rank block an extension
    # touches + 1 B per LCS element a contraction reads + 1 B of query in (the MS values stay on the device; sites leave) -, the op
    # counts by the oracle on a sample of the same reads; the walk kernel's duration from the events around every launch
    roofline = cpu = None
    if not args.no_cpu_baseline:
        from oracle import binding as ora
        n_s = max(1, min(args.reads, int(20_000_000 // args.read_len)))  # ~20 Mbases of the timed reads
        cn = ora.Counters()
        t1 = time.perf_counter()
        oi.matches_batch(concat[:n_s * args.read_len], offsets[:n_s + 1], 1e-7, n_threads=cores, counters=cn)
        c = cn.as_dict()
        sb = n_s * args.read_len
        b_alg = (64.0 * c.get("rank_blocks", 0) + c.get("lcs_reads", 0)) / sb + 1.0
        ach = b_alg * bases / (walk_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                    "traffic": None, "algorithmic_bytes_per_base": round(b_alg, 2), "units_per_launch": bases,
                    "kernel": "the call mode of the walk (plan_kernel + guided walk over recovery lines, or ms_walk_kernel<CALL>): MS values + breakpoint scan",
                    "kernel_ms": round(walk_ms, 4), "counted_on": f"{n_s} of the timed reads by the oracle (SURVEY.md 8(d))",
                    "note": "no depth table at this index size (17 bases would be present by chance): the walk is the round-3 route; "
                            "the one kernel for sequences of any length (long_kernels.hip) needs a table",
                    "frac_meaning": "SURVEY.md 8(d)'s bytes of the REFERENCE algorithm (64 B per rank block an extension of its walk touches, counted by the "
                                    "oracle) over the walk's duration and the HBM peak: the plan-guided walk skips most of those extensions (bases on a "
                                    "diagonal cost a comparison with the text), so this is how fast the reference's work is disposed of, not the share of "
                                    "the HBM bandwidth the kernels use - that takes the PMC passes (`traffic`), not taken at this size"}
        # ---- CPU baseline: the oracle's literal kbo::call (per-sequence index build + both passes) on a bounded sample, one thread a read
        n_c = min(args.reads, 64)
        t1 = time.perf_counter()
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(lambda s_: oi.call(concat[int(offsets[s_]):int(offsets[s_ + 1])].tobytes(), args.k, 1e-7), range(n_c)))
        dt = time.perf_counter() - t1
        cpu = {"value": round(n_c * args.read_len / dt / 1e6, 2), "unit": "Mbp/s", "cores": cores, "kind": "port",
               "sample": f"oracle.call (kbo::call, lib.rs:547-573) on the first {n_c} reads, {cores} threads, one read each at a time"}
    print(json.dumps({
        "metric": (f"query Mbp/sec for kbo call, k={args.k}, {args.genome / 1e6:g} Mbp SBWT, {args.read_len} bp reads (first pass device-resident; "
                   "whole_call: host sequences in, variants out)") if args.c5 else
                  f"query Mbp/sec for kbo call first pass (MS walk whose lanes run the breakpoint scan; sites only leave the device), k={args.k}, "
                  f"{args.genome / 1e6:g} Mbp SBWT",
        "roofline": roofline, "cpu_baseline": cpu,
        "value": round(bases * args.steps / elapsed / 1e6, 1), "unit": "Mbp/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": ("C5, one GPU's share of 8: " if args.c5 else "C5 shape, scaled: ") + f"kbo call first pass, {args.genome / 1e6:g} Mbp iid genome SBWT k={args.k}, "
                               f"{args.reads} x {args.read_len} bp reads, {args.sub_rate * 100:g}% substitutions",
                   "threshold": thr, "sites_per_step": n_sites, "bytes_leaving_the_device_per_base": round(16 * n_sites / bases, 4)},
        "kernels_ms": {"ms_walk_call_mode": round(walk_ms, 4)},
        "whole_call": whole,
        "bit_exact_vs_oracle": exact, "parity_scope": "sites of every read vs the oracle's first pass of call_variants"}), flush=True)


