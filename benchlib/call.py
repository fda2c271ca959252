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


def _phase_kernels_of_a_call(fn):
    """the phases of one kbo_call_batch_flat call as the library itself times them (KBO_TIMING=1: stderr lines of call_batch.cpp) ->
    {phase: ms}.  The library's stderr is the process's file descriptor 2: captured through a pipe for the one call."""
    import re
    import tempfile
    sys.stderr.flush()
    saved = os.dup(2)
    got = {}
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        os.dup2(tmp.fileno(), 2)
        os.environ["KBO_TIMING"] = "1"
        try:
            out = fn()
        finally:
            os.environ.pop("KBO_TIMING", None)
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        txt = tmp.read().decode(errors="replace")
    m = re.search(r"(\d+) slabs: staging ([\d.]+) ms, enqueue ([\d.]+), waiting for the device ([\d.]+), taking results ([\d.]+), slow route ([\d.]+) \((\d+) slabs\); (\d+) sites", txt)
    if m:
        got = {"slabs": int(m.group(1)), "staging_copies_ms": float(m.group(2)), "enqueue_ms": float(m.group(3)), "waiting_for_the_device_ms": float(m.group(4)),
               "taking_results_ms": float(m.group(5)), "slow_route_ms": float(m.group(6)), "slabs_on_the_slow_route": int(m.group(7)),
               "sites_resolved_on_the_host": int(m.group(8))}
    m = re.search(r"device passes \+ results\s+([\d.]+) ms", txt)
    if m:
        got["device_passes_and_results_ms"] = float(m.group(1))
    return out, got


def main_call(args):
    """kbo call, first pass (variant_calling.rs:266-273) over a batch of long reads resident in HBM: the walk in call mode
    (its lanes run the breakpoint scan); what leaves the device is one 16-byte record per site.  Parity: the sites of
    every read against the oracle's first pass.  Then the whole call (kbo_call_batch_flat: host sequences in, variants out)."""
    import torch
    import kbo_amd
    from kbo_amd import batch, derandomize, synth
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cores, _ = usable_cores()
    t_all = time.perf_counter()
    phases = {}

    def lap(name, t0):
        phases[name] = round(time.perf_counter() - t0, 2)
        print(f"[bench C5 t={time.perf_counter() - t_all:7.1f} s] {name}: {phases[name]} s", file=sys.stderr, flush=True)

    t0 = time.perf_counter()
    genome, sbwt = build_or_load_index(args, cores)
    lap("genome + index (host build or cache file)", t0)
    t0 = time.perf_counter()
    concat, offsets = synth.reads(genome, args.reads, args.read_len, args.sub_rate)
    lap("reads", t0)
    t0 = time.perf_counter()
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=device)
    lap("device copy of the index (layout, plan structures, upload) + the batch", t0)
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
    # the walk's own work, counted by its kernels on the timed reads (one more launch with the counters on: kbo_set_plan_stats)
    own = None
    L.kbo_set_plan_stats(1)
    try:
        step()
        torch.cuda.synchronize(device)
        own = dev.plan_stats()
    except kbo_amd.KboError:
        own = None
    finally:
        L.kbo_set_plan_stats(0)
    counts_all = count.cpu().numpy()
    counts = counts_all[:LISTS * 16:16]
    n_sites, seg = int(counts.sum()), cap // LISTS
    fits = bool((counts <= seg).all()) and int(counts_all[LISTS * 16]) == 0
    sites_h = sites.cpu().numpy().view(np.uint32)
    raw = np.concatenate([sites_h[g * seg:g * seg + min(int(counts[g]), seg)] for g in range(LISTS)]) if n_sites else sites_h[:0]
    raw = raw[raw[:, 0] != 0xFFFFFFFF]  # (void records: kbo_hip.h, kbo_call_walk_dev)
    n_sites = len(raw)
    exact = None
    oi = None
    if not args.no_cpu_baseline:  # every read: the oracle's first pass of call_variants (ora_call_sites_batch)
        from oracle import binding as ora
        t0 = time.perf_counter()
        rows, Carr, lcs = sbwt.export_parts()
        oi = ora.Index.from_parts(args.k, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
        lap("oracle: adopting the index's parts", t0)
        t0 = time.perf_counter()
        recs = oi.call_sites_batch(concat, offsets, thr, n_threads=cores)
        base = offsets[recs[:, 0].astype(np.int64)]
        want = np.stack([base + recs[:, 1], base + recs[:, 2], recs[:, 3]], axis=1).astype(np.uint64)
        got = raw[:, :3].astype(np.uint64)
        want = want[np.lexsort(want.T[::-1])]
        got = got[np.lexsort(got.T[::-1])]
        exact = bool(fits and want.shape == got.shape and np.array_equal(want, got))
        lap("oracle: the first pass of every read + comparison", t0)
    # the whole of kbo::call over the same reads through the product entry point (host sequences in, variants out): both passes and the
    # variants' order, case analysis and characters on the device (call_second_kernels.hip, call_emit_kernels.hip), two slots on two
    # streams; a sample against the oracle's literal call
    whole = None
    if not args.no_whole_call:
        try:
            t0 = time.perf_counter()
            opts = kbo_amd.CallOpts(sbwt_build_opts=kbo_amd.BuildOpts(k=args.k, build_select=True))
            best, res, ph = 1e9, None, {}
            for it in range(3):
                t1 = time.perf_counter()
                if it == 2:  # (the third call with the library's own phase clock on)
                    res, ph = _phase_kernels_of_a_call(lambda: batch.call_batch_arrays(sbwt, concat, offsets, opts))
                else:
                    res = batch.call_batch_arrays(sbwt, concat, offsets, opts)
                best = min(best, time.perf_counter() - t1)
            whole = {"entry_point": "kbo_call_batch_flat", "ms": round(best * 1e3, 2), "us_per_read": round(best / args.reads * 1e6, 2),
                     "mbp_per_s": round(dev.total / best / 1e6, 1), "variants": int(res["var_offsets"][-1]),
                     "phases_of_one_call_host_clock_ms": ph,
                     "note": "host sequences (pageable) in, variants out as flat arrays in the order of (sequence, query position): best of 3 calls; "
                             "the phases are the calling thread's own clock in one call - staging copies into pinned memory, enqueueing a slab's "
                             "launches, waiting for the device, copying results out of pinned memory - the device works beside all but the waiting"}
            # the same through kbo_call_batch (the reference's records, two pointers per variant, made from the flat arrays by host threads)
            t1 = time.perf_counter()
            vo = np.zeros(args.reads + 1, dtype=np.uint64)
            import ctypes as C
            from kbo_amd import _capi
            co = _capi.CallOpts(opts.max_error_prob, opts.sbwt_build_opts._to_c())
            pv = C.POINTER(_capi.Variant)()
            kbo_amd.check(L.kbo_call_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, args.reads, C.byref(co), C.byref(pv), vo.ctypes.data))
            whole["kbo_call_batch_records_ms"] = round((time.perf_counter() - t1) * 1e3, 2)
            same = int(vo[-1]) == int(res["var_offsets"][-1]) and np.array_equal(vo, res["var_offsets"])
            L.kbo_free(pv)
            whole["kbo_call_batch_same_offsets"] = bool(same)
            lap("whole call x 4", t0)
            if oi is not None:
                t0 = time.perf_counter()
                rng = np.random.default_rng(1)
                pick = [int(x) for x in rng.integers(0, args.reads, min(40, args.reads))]
                ok = True
                for s_ in pick:
                    a_, b_ = int(offsets[s_]), int(offsets[s_ + 1])
                    exp_calls, _, _ = oi.call(concat[a_:b_].tobytes(), args.k, 1e-7)
                    ok = ok and [(p_, q_.decode(), r_.decode()) for p_, q_, r_ in batch.variants_of(res, s_)] == exp_calls
                whole["equal_to_oracle_call_on_sampled_reads"] = len(pick) if ok else False
                lap("oracle.call on 40 sampled reads", t0)
        except kbo_amd.KboError as e:  # (a threshold the reference refuses, a sharded index: said, not hidden)
            whole = {"error": str(e)}
    bases = dev.total
    # ---- roofline of the first pass.  `frac` = the walk's OWN compulsory bytes - what its kernels have to move for the work they
    # counted themselves on the timed reads - over the walk's duration and the HBM peak:
    #   per base      1 B query in (staged once, whole lines) + 1 B MS value out (kbo_call_walk_dev leaves the values on the device for the
    #                 second pass) + 1 B of path-cover text on the diagonal (plan_kernel's byte compare)
    #   per item      16 B (its record)                      per seed look-up   8 B interval + 4 B text position
    #   per seed extension beyond the table  2 x 16 B rank blocks
    #   per unit      2 x 16 B (written by plan_emit, read by the walk)
    #   per walked base (accepted / failed extension, entry level)   2 x 32 B of the two recovery lines (rank block + LCS window
    #                 of l's and r's 64-row block, sbwt_index.hpp; 2 x 16 B rank blocks on indexes that walk without them)
    #   per site      16 B out
    # Beside it, as before: SURVEY.md 8(d)'s bytes of the REFERENCE algorithm for the same reads (counted by the oracle) over the
    # same duration - how fast the reference's work is disposed of, not a share of the bandwidth.
    roofline = cpu = None
    own_bytes = None
    if own:
        walked = own["accepted"] + own["failed"] + own["entry_levels"]
        line_b = 64 if sbwt.device_layout().get("lines_bytes", 0) else 32
        by_part = {"query_in": bases, "ms_values_out": bases, "text_on_the_diagonal": bases,
                   "seed_lookups": 12 * own["seed_lookups"], "seed_extensions": 32 * own["seed_extensions"],
                   "units": 32 * own["units"], "walked_bases": line_b * walked, "sites_out": 16 * n_sites}
        own_bytes = int(sum(by_part.values()))
    wl_key = f"{args.genome}x{args.reads}x{args.read_len}x{args.sub_rate:g}:call"
    traffic = tsrc = tmiss = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            entry = json.load(open(tpath)).get("workloads", {}).get(wl_key)
            if entry:
                traffic, tmiss, tsrc = entry.get("a1_bytes_per_launch"), entry.get("a1_tcc_miss_per_launch"), entry.get("source")
                if entry.get("build_sha16") != build_sha16():
                    tsrc = f"{tsrc} (taken of build {entry.get('build_sha16')}, this one is {build_sha16()}: the walk kernels' sources may differ)"
        except Exception:
            pass
    if own_bytes:
        ach = own_bytes / (walk_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                    "traffic": traffic, "traffic_source": tsrc,
                    "wasted_traffic": round(traffic / own_bytes, 2) if traffic else None,
                    "traffic_frac": round(traffic / (walk_ms * 1e-3) / 8e12, 4) if traffic else None,
                    "l2_miss_per_launch": tmiss,
                    "algorithmic_bytes_per_base": round(own_bytes / bases, 3), "units_per_launch": bases,
                    "bytes_by_part": by_part, "walk_counters": own,
                    "kernel": "the call mode of the plan-guided walk (plan_kernel, plan_count / scan / plan_emit, ms_walk_guided_kernel or "
                              "ms_walk_recovery_kernel <CALL>, redo_collect + ms_walk_kernel, call_fix_sites): MS values + breakpoint scan",
                    "kernel_ms": round(walk_ms, 4),
                    "frac_meaning": "the walk's OWN compulsory bytes (bytes_by_part: counted by its kernels on the timed reads, priced as "
                                    "benchlib/call.py says) x 1 launch / the walk's duration (HIP events on its stream around every launch) / 8 TB/s.  "
                                    "A chain of dependent look-ups per unit - 64 B used of every 128 B line it fetches - on an index far beyond any cache: "
                                    "bound by memory LATENCY at the waves the kernel keeps resident, not by bytes (DESIGN.md section 4.4)",
                    "note": "no depth table at this index size (17 bases would be present by chance), so neither direct-form kernel applies: "
                            "the walk is the round-3 route"}
    if oi is not None:
        from oracle import binding as ora
        t0 = time.perf_counter()
        n_s = max(1, min(args.reads, int(20_000_000 // args.read_len)))  # ~20 Mbases of the timed reads
        cn = ora.Counters()
        oi.matches_batch(concat[:n_s * args.read_len], offsets[:n_s + 1], 1e-7, n_threads=cores, counters=cn)
        c = cn.as_dict()
        sb = n_s * args.read_len
        b_alg = (64.0 * c.get("rank_blocks", 0) + c.get("lcs_reads", 0)) / sb + 1.0
        ref_ach = b_alg * bases / (walk_ms * 1e-3) / 1e9
        if roofline is None:
            roofline = {"bound": "hbm", "achieved": round(ref_ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ref_ach / 8000.0, 4), "traffic": None,
                        "algorithmic_bytes_per_base": round(b_alg, 2), "units_per_launch": bases, "kernel_ms": round(walk_ms, 4),
                        "frac_meaning": "SURVEY.md 8(d)'s bytes of the reference algorithm (the walk's own counters were not available)"}
        roofline["reference_algorithm_bytes_per_base"] = round(b_alg, 2)
        roofline["frac_reference_algorithm"] = round(ref_ach / 8000.0, 4)
        roofline["reference_algorithm_counted_on"] = f"{n_s} of the timed reads by the oracle (SURVEY.md 8(d)): the plan-guided walk skips most of those extensions"
        lap("oracle: op counts on a sample", t0)
        # ---- CPU baseline: the oracle's literal kbo::call (per-sequence index build + both passes) on a bounded sample, one thread a read
        t0 = time.perf_counter()
        n_c = min(args.reads, 64)
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(lambda s_: oi.call(concat[int(offsets[s_]):int(offsets[s_ + 1])].tobytes(), args.k, 1e-7), range(n_c)))
        dt = time.perf_counter() - t0
        cpu = {"value": round(n_c * args.read_len / dt / 1e6, 2), "unit": "Mbp/s", "cores": cores, "kind": "port",
               "sample": f"oracle.call (kbo::call, lib.rs:547-573) on the first {n_c} reads, {cores} threads, one read each at a time"}
        lap("oracle: cpu baseline", t0)
    lay = sbwt.device_layout()
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
                   "threshold": thr, "sites_per_step": n_sites, "bytes_leaving_the_device_per_base": round(16 * n_sites / bases, 4),
                   "setup_seconds": {k_: round(float(v_), 2) for k_, v_ in lay.items() if k_.endswith("_seconds")},
                   "index_device_bytes": {k_: int(v_) for k_, v_ in lay.items() if k_.endswith("_bytes")},
                   "host_phases_seconds": phases, "build_sha16": build_sha16()},
        "kernels_ms": {"ms_walk_call_mode": round(walk_ms, 4)},
        "whole_call": whole,
        "bit_exact_vs_oracle": exact, "parity_scope": "sites of every read vs the oracle's first pass of call_variants"}), flush=True)
