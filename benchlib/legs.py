"""bench.py: what runs behind the timed region on rank 0 - the oracle (test infrastructure: parity gate and CPU baseline only), the
variants of the workload, the entry points that return matching statistics, host buffers in and out."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403


def cpu_baseline_leg(args, oi, concat, offsets, gpu_d, gpu_chars):
    """Times the oracle (C restatement of the reference algorithm, sbwt-like layout) on a bounded sample of the same reads
    with all host cores, checks the GPU output against it, and returns (cpu_baseline dict, B_ref bytes/base, bit_exact, ops)."""
    from oracle import binding as ora
    cores, cores_note = usable_cores()
    L = args.read_len
    n_all = len(offsets) - 1
    # calibration slice (also warms the index), then a sample sized to the time budget, walked by a pinned thread pool after
    # an untimed warm-up pass (oracle/kbo_oracle.c ora_matches_batch_timed: outputs allocated and touched beforehand, reads
    # handed out dynamically).  Three timed runs of `passes` passes each: the MEDIAN is quoted (one run is noisy on a
    # shared box: round 2's driver saw 696 Mbp/s where the builder saw 453 - 688).
    n0 = min(n_all, 20_000)
    _, _, dt0 = oi.matches_batch_timed(concat[:n0 * L], offsets[:n0 + 1], 1e-7, n_threads=cores, passes=1)
    dt0 = max(dt0, 1e-4)
    budget = 0.6 * args.cpu_seconds / 3.0
    n1 = int(min(n_all, max(n0, n0 * budget / dt0)))
    passes = int(max(1, min(50, budget / max(dt0 * n1 / n0, 1e-3))))
    rates, sec_all = [], 0.0
    chars = d = None
    for _ in range(3):
        chars, d, sec = oi.matches_batch_timed(concat[:n1 * L], offsets[:n1 + 1], 1e-7, n_threads=cores, passes=passes)
        rates.append(n1 * L * passes / sec / 1e6)
        sec_all += sec
    allcore = float(np.median(rates))
    # operation counts of the reference algorithm (separate, untimed, counted run)
    ctr = ora.Counters()
    nc = min(n1, 50_000)
    oi.matches_batch(concat[:nc * L], offsets[:nc + 1], 1e-7, n_threads=cores, counters=ctr)
    c = ctr.as_dict()
    b_ref = (64.0 * c["rank_blocks"] + 1.0 * c["lcs_reads"]) / c["bases"] + 2.0
    exact = bool(np.array_equal(d, gpu_d[:n1 * L]) and np.array_equal(chars, gpu_chars[:n1 * L]))
    # single-thread rate of the same restatement (SURVEY.md section 8(d) asks for both): same driver, ~ a quarter of the budget,
    # median of three runs as well
    t1 = 0.3 * args.cpu_seconds / 3.0
    ns = int(max(2_000, min(n1, n0 * t1 / (dt0 * cores))))
    singles = []
    for _ in range(3):
        _, _, sec1 = oi.matches_batch_timed(concat[:ns * L], offsets[:ns + 1], 1e-7, n_threads=1, passes=1, want_d=False)
        singles.append(ns * L / max(sec1, 1e-6) / 1e6)
    single = float(np.median(singles))
    base = {"value": round(allcore, 3), "unit": "Mbp/s", "cores": cores, "kind": "port",
            "runs_mbps": [round(r, 1) for r in rates],
            "single_thread_value": round(single, 3), "single_thread_runs_mbps": [round(r, 1) for r in singles],
            "scaling_efficiency": round(allcore / max(single * cores, 1e-9), 3), "cores_note": cores_note,
            "sample": f"first {n1} of the {n_all} reads ({n1 * L / 1e6:.1f} Mbp), median of 3 runs of {passes} timed passes each after a "
                      f"warm-up pass, oracle/kbo_oracle.c ora_matches_batch_timed on a pool of {cores} pinned threads, "
                      f"{sec_all:.1f} s wall ({sec_all * cores:.0f} core-seconds); single thread: {ns} reads, median of 3"}
    ops = {k: round(v / c["bases"], 4) for k, v in c.items() if k != "bases"}
    return base, b_ref, exact, ops


def stage_model_leg(args, sbwt, oi, concat, offsets, gpu_d, n_sample=2_000_000):
    """The CPU model of the plan-guided stage (oracle/plan_model.c, pinned to the kernels' own counters by
    tests/test_gpu_model.py) over the timed reads: its MS values are checked against the GPU's, its work counts give the
    stage's compulsory bytes per base (B_plan) and the distinct 128-byte lines a unit touches."""
    from oracle import binding as ora
    cores, _ = usable_cores()
    n = min(len(offsets) - 1, n_sample)
    L = args.read_len
    order = sbwt.depth_table_order()
    P = ora.shipped_plan_params(args.k, sbwt.n_sets(), depth_table=order)
    ms, cn = oi.plan_model(sbwt.path_cover(), P, concat[:n * L], offsets[:n + 1], n_threads=cores)
    same = bool(np.array_equal(ms, gpu_d[:n * L]))
    iters = cn["walk_accepted"] + cn["walk_failed"] + cn["walk_contractions"]
    by = {
        # plan_kernel's streams: the queries, the text of their diagonals, the predicted MS values
        "streams": cn["bases"] + 2 * cn["compare_bases"],
        "seeds": 8 * cn["seed_lookups"] + 32 * cn["seed_extensions"] + 4 * cn["pos_lookups"],
        "redo": 16 * cn["items_flagged"] + 32 * cn["redo_iterations"] + 2 * cn["redo_bases"],
    }
    if order:
        # item records: WalkItem read, GuidedItem written and read by the resolve kernel, redo flag written and read,
        # mismatch lists written once and read once
        by["item_records"] = cn["items"] * (16 + 16 + 16 + 2) + 4 * max(0, cn["mismatches"] - cn["items"] + cn["items_unseeded"])
        # the table: one byte per look-up; the bases in front of and behind a mismatch that its look-ups are keyed by
        # (order + 1 lanes, 32 bases each, overlapping: order + 32 bytes); the values written
        by["table"] = cn["tab_lookups"] + (order + 32) * cn["mismatches"] + cn["tab_written"]
    else:
        # item records: WalkItem read, GuidedItem written and read by count + emit, redo flag, unit counts through the scan,
        # mismatch lists written once and read twice
        by["item_records"] = cn["items"] * (16 + 16 + 32 + 1 + 8 + 16) + 6 * max(0, cn["mismatches"] - cn["items"] + cn["items_unseeded"])
        # a unit: record written and read, start row, two query blocks, its output bytes
        by["unit_records"] = cn["units"] * (32 + 32 + 32) + 4 * cn["node_lookups"] + cn["walk_out_bytes"]
        # the walk: two 16-byte loads per iteration (rank blocks or entries); over recovery lines two rank blocks + two LCS
        # windows per iteration, two entries per level taken from the entries
        by["walk"] = (64 * cn["walk_iterations_lines"] + 32 * cn["walk_entry_levels"]) if P.recovery_lines else 32 * iters
    total = float(sum(by.values()))
    per_base = {k: round(v / cn["bases"], 4) for k, v in by.items()}
    units = max(1, cn["units"])
    summary = {
        "sample_reads": n, "ms_equal_to_gpu": same, "gave_up": bool(cn["gave_up"]),
        "form": (f"depth table of {order} bases" if order else "recovery lines" if P.recovery_lines else "rank blocks + entries"),
        "parameters": {"seed_table_depth": P.seed_table_depth, "seed_depth": P.seed_depth, "gap": P.gap, "chunk": P.chunk,
                       "list_cap": P.list_cap, "bail_x16": P.bail_x16, "depth_table": order},
        "unseeded_reads": cn["items_unseeded"], "flagged_reads": cn["items_flagged"],
        "seed_extensions_per_read": round(cn["seed_extensions"] / cn["items"], 3),
        "mismatches_per_read": round(cn["mismatches"] / cn["items"], 4),
        "bytes_per_base": per_base,
    }
    if order:
        mm = max(1, cn["mismatches"])
        summary["per_mismatch"] = {"table_lookups": round(cn["tab_lookups"] / mm, 3), "values_written": round(cn["tab_written"] / mm, 3)}
        st = max(1, cn["tab_stretches"])  # (the mismatches of reads on a wrong diagonal are not looked up)
        summary["per_stretch"] = {"table_lookups": round(cn["tab_lookups"] / st, 3), "values_written": round(cn["tab_written"] / st, 3)}
        # lines that cannot come from L2: the streams (query, text of the diagonal, MS: 3 x read length / 128), a seed-table entry
        # per look-up and the seed's text position, and the table: the look-ups of a mismatch are consecutive bases, three of
        # which share a 64-byte line (a run of P bases touches (P + 2) / 3 of them)
        summary["fills_min_per_read"] = round(3 * L / 128 + (cn["seed_lookups"] + cn["pos_lookups"]) / cn["items"]
                                              + (cn["tab_lookups"] + 2 * cn["tab_stretches"]) / 3 / cn["items"], 3)
        summary["stretches_per_read"] = round(cn["tab_stretches"] / cn["items"], 4)
    else:
        summary["units_per_read"] = round(cn["units"] / cn["items"], 4)
        summary["per_unit"] = {"accepted": round(cn["walk_accepted"] / units, 3), "failed": round(cn["walk_failed"] / units, 3),
                               "contraction_levels": round(cn["walk_contractions"] / units, 3),
                               "entry_levels": round(cn["walk_entry_levels"] / units, 3),
                               "iterations": round((cn["walk_iterations_lines"] if P.recovery_lines else iters) / units, 3),
                               # 128-byte lines one unit touches (record, start row, query, output, index); "beyond_l2" leaves out the
                               # rank blocks when all of them fit one XCD's 4 MiB L2 (C2: 3.3 MB - they stay resident, shared by all units)
                               "distinct_lines": round(cn["unit_distinct_lines"] / units, 3),
                               "distinct_lines_beyond_l2": round((cn["unit_distinct_lines"] - (cn["unit_distinct_rank_lines"]
                                                                  if sbwt.device_bytes()[0] < (4 << 20) else 0)) / units, 3)}
    return total / cn["bases"], summary, cn



def sensitivity_leg(args, genome, sbwt, oi, torch, device, stream, pipes=None):
    """SURVEY.md 8(d) asks for 0 % and 5 % variants of C2; VERDICT adds what the iid forward reads hide: reads from the
    other strand (the index has no reverse complements), reads from elsewhere, a repeat-rich genome of many contigs.  Each:
    a resident batch of the C2 shape, 8 warm-up + 40 timed steps (four sets of buffers in flight on two pipelines as in the headline), every one of its first 20 000 reads against the oracle."""
    import kbo_amd
    from kbo_amd import batch, synth
    from oracle import binding as ora
    cores, _ = usable_cores()
    n_reads, L = min(args.reads, 1_000_000), args.read_len
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b

    def measure(name, ix, o, concat, offsets, note):
        dev = batch.DeviceBatch(ix, concat, offsets, device=device, format=True, want_ms=False)
        devs = [dev]
        if pipes is not None:  # (in flight as in the headline: the same reads, further sets of buffers)
            for _ in range(2 * pipes - 1):
                devs.append(batch.DeviceBatch(ix, concat, offsets, device=device, format=True, want_ms=False))
        elapsed, a1, dt, _ = run_batch(devs, stream, False, 40, 8, torch, device, args.two_kernels, pipes if len(devs) > 1 else None)
        fused = dev.fused
        n_all = len(offsets) - 1
        n_chk = max(1, min(n_all, int(np.searchsorted(offsets, 3_000_000))))  # the reads of the first 3 Mbp
        n_b = int(offsets[n_chk])
        exp_chars, exp_d = o.matches_batch(concat[:n_b], offsets[:n_chk + 1], 1e-7, n_threads=cores, want_d=True)
        exp_map = np.frombuffer(ora.relative_to_ref(concat[:n_b], exp_chars), dtype=np.uint8)
        ok = bool(np.array_equal(dev.chars[:n_b].cpu().numpy(), exp_map))  # (what the timed steps left behind)
        extra = {}
        if fused and dev.max_len > 160:  # (sequences of any length: what the kernel's pieces did - one more call, over the batch's own work buffer)
            dev.run(stream)
            st = dev.long_stats(stream)
            extra = {"pieces": st["pieces"], "flagged_pieces": st["flagged"]}  # (flagged: to the plain walk + the literal recurrences)
        dev.walk(stream)
        torch.cuda.synchronize(device)
        ok = bool(ok and np.array_equal(dev.ms[:n_b].cpu().numpy(), exp_d))
        del dev, devs
        return {**extra, "variant": name, "value": round(int(offsets[-1]) * 40 / elapsed / 1e6, 1), "unit": "Mbp/s", "steps": 40, "one_kernel": fused,
                "step_ms": round(a1 + dt, 4), "bit_exact_vs_oracle": ok, "note": note}

    out = []
    L_ = kbo_amd.lib()
    for sub in (0.0, 0.05):
        L_.kbo_set_plan(1, 0, 0)  # (every variant starts with a clean hold-off)
        concat, offsets = synth.reads(genome, n_reads, L, sub, seed=0x5E115 + int(sub * 1000))
        out.append(measure(f"{sub * 100:g}% substitutions", sbwt, oi, concat, offsets,
                           ("7.5 mismatches per read against the diagonal: with the depth table each costs its look-ups; with units (larger "
                            "indexes) the stage gives the plan up above ~4 % and walks plainly") if sub > 0.04 else
                           "error-free: plan_kernel alone, nothing behind it"))
    # insertions and deletions: 1 % substitutions + 0.2 % of the bases start an insertion or a deletion of 1 - 3
    # bases (a quarter of the reads have one).  Such a read leaves its diagonal: the kernel cuts it between two diagonals
    L_.kbo_set_plan(1, 0, 0)
    concat, offsets = indel_reads(genome, n_reads, L, 0.01, 0.002, seed=0x5E11C)
    out.append(measure("1% substitutions + 0.2% insertions / deletions", sbwt, oi, concat, offsets,
                       "a read with an insertion or a deletion follows two diagonals of the text: the kernel seeds the second from the "
                       "read's last bases and cuts the read where the two together mismatch least"))
    # sequences of more than 160 bases - what kbo::map / find / call are called with (lib.rs:612-628, 720-761): one wave per piece
    # of a sequence (long_kernels.hip), the pieces whose proof fails by the plain walk + the literal recurrences behind it
    L_.kbo_set_plan(1, 0, 0)
    n_long = max(100, n_reads * L // 10_000)
    concat, offsets = synth.reads(genome, n_long, 10_000, 0.01, seed=0x5E11E)
    out.append(measure("10 kbp reads, 1% substitutions", sbwt, oi, concat, offsets,
                       "pieces of 945 own bases inside regions of 1008; a seed per piece, the text on its diagonal staged in LDS"))
    # ONT-like reads (C5's premise): 10 kbp, 5 % errors of which half are insertions / deletions
    L_.kbo_set_plan(1, 0, 0)
    n_long = max(100, n_reads * L // 10_000 // 2)
    concat, offsets = indel_reads(genome, n_long, 10_000, 0.025, 0.025 / 2, seed=0x5E11D, many=True)
    out.append(measure("ONT-like: 10 kbp reads, 2.5% substitutions + 2.5% insertions / deletions", sbwt, oi, concat, offsets,
                       "a diagonal is lost every 80 bases: the 64 lanes of the piece's wave try the 64 diagonals beside it"))
    L_.kbo_set_plan(1, 0, 0)
    concat, offsets = synth.reads(genome, n_reads, L, 0.01, seed=0x5E117)
    rc = comp[concat.reshape(-1, L)[:, ::-1]].reshape(-1).copy()
    out.append(measure("reverse-strand reads, 1% substitutions", sbwt, oi, rc, offsets,
                       "the index holds one strand (BuildOpts::add_revcomp=false, the crate default): nothing seeds, plain walk"))
    L_.kbo_set_plan(1, 0, 0)
    other = synth.genome(args.genome, seed=0xBADC0DE)
    concat, offsets = synth.reads(other, n_reads, L, 0.0, seed=0x5E118)
    out.append(measure("unrelated reads", sbwt, oi, concat, offsets, "reads of another random genome: MS values of 10 - 13 everywhere"))
    del other
    # a repeat-rich genome of many contigs: 40 contigs, a fifth of every contig copied from elsewhere, short tandem arrays
    L_.kbo_set_plan(1, 0, 0)
    rng = np.random.default_rng(0x5E119)
    base = synth.genome(args.genome, seed=0x5E11A)
    contigs = []
    clen = max(2 * L, args.genome // 40)
    for c in range(40):
        piece = base[c * clen:(c + 1) * clen].copy()
        if len(piece) < 2 * L:
            break
        for _ in range(8):  # copies of 2.5 % of the contig from anywhere in the genome
            n = max(L, clen // 40)
            src = int(rng.integers(0, len(base) - n))
            dst = int(rng.integers(0, len(piece) - n))
            piece[dst:dst + n] = base[src:src + n]
        t0 = int(rng.integers(0, len(piece) - 2000))
        piece[t0:t0 + 2000] = np.tile(piece[t0:t0 + 50], 40)  # a tandem array
        contigs.append(piece)
    rix, _ = kbo_amd.build(contigs, kbo_amd.BuildOpts(k=args.k, num_threads=min(16, cores)))
    rows, Carr, lcs = rix.export_parts()
    roi = ora.Index.from_parts(args.k, rix.n_sets(), rix.n_kmers(), rows, Carr, lcs)
    cat = np.concatenate(contigs)
    concat, offsets = synth.reads(cat, n_reads, L, 0.01, seed=0x5E11B)
    e = measure("repeat-rich genome, 40 contigs", rix, roi, concat, offsets,
                "a fifth of every contig duplicated from elsewhere + tandem arrays: path cover of many paths; reads that cross a path "
                "start or leave their diagonal in a repeat go to the redo pass")
    e["index_n_sets"] = rix.n_sets()
    out.append(e)
    L_.kbo_set_plan(1, 0, 0)
    return out


def ms_leg(args, sbwt, oi, concat, offsets, torch, device, stream, pipes):
    """The forms of the API that return the MATCHING STATISTICS (index.rs:243-256; `metric` says "bit-exact MS"): kbo_ms_batch_dev (MS
    bytes only) and kbo_map_batch_dev with want_ms (map_reads_kernel's MS-emitting instantiation: MS bytes + characters), each timed
    over the headline's batch - resident, 8 warm-up + 40 timed steps (the walk: 4 + 12), in flight like the headline where the entry point has a tail
    stream - and every MS byte (and character) of the batch compared with the oracle."""
    from kbo_amd import batch
    from oracle import binding as ora
    cores, _ = usable_cores()
    exp_chars, exp_d = oi.matches_batch(concat, offsets, 1e-7, n_threads=cores, want_d=True)
    exp_map = np.frombuffer(ora.relative_to_ref(concat, exp_chars), dtype=np.uint8)
    total = int(offsets[-1])
    out = {}
    # kbo_ms_batch_dev: MS bytes out and nothing else - for reads over a copy with a depth table map_reads_kernel's MS-emitting form stopped
    # behind the values + the plain walk of the reads it leaves (one stream: the entry point has no tail stream)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=device, format=True, want_ms=True)
    def warm(n):
        for _ in range(n):
            dev.walk(stream)
    condition(warm, 4, torch, device)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(12):
        dev.walk(stream)
    e1.record(stream)
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / 12
    out["kbo_ms_batch_dev"] = {"value": round(total / ms / 1e3, 1), "unit": "Mbp/s", "step_ms": round(ms, 4), "bytes_out_per_base": 1,
                               "bit_exact_vs_oracle": bool(np.array_equal(dev.ms[:total].cpu().numpy(), exp_d))}
    # ... and the caller's way to keep the device full with it: resident batches in turn on two streams of its own (a call is
    # asynchronous on its stream; its second pass then runs beside the other stream's kernel)
    devs = [dev] + [batch.DeviceBatch(sbwt, concat, offsets, device=device, format=True, want_ms=True) for _ in range(3)]
    streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)]

    def go(n):
        for i in range(n):
            devs[i % 4].walk(streams[i % 2])
    condition(go, 8, torch, device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    go(40)
    torch.cuda.synchronize(device)
    el = time.perf_counter() - t0
    out["kbo_ms_batch_dev"]["two_streams"] = {"value": round(total * 40 / el / 1e6, 1), "unit": "Mbp/s", "steps": 40, "step_ms": round(el / 40 * 1e3, 4),
                                              "batches_in_flight": 4, "bit_exact_vs_oracle": all(bool(np.array_equal(d.ms[:total].cpu().numpy(), exp_d)) for d in devs)}
    del dev, devs
    # kbo_map_batch_dev(want_ms): the one kernel in its MS-emitting form, MS bytes + formatted characters out
    devs = [batch.DeviceBatch(sbwt, concat, offsets, device=device, format=True, want_ms=True) for _ in range(2 * pipes if pipes else 1)]
    elapsed, _, _, _ = run_batch(devs, stream, False, 40, 8, torch, device, False, pipes)
    ok = all(bool(np.array_equal(d.ms[:total].cpu().numpy(), exp_d) and np.array_equal(d.chars[:total].cpu().numpy(), exp_map)) for d in devs)
    out["kbo_map_batch_dev_want_ms"] = {"value": round(total * 40 / elapsed / 1e6, 1), "unit": "Mbp/s", "steps": 40, "step_ms": round(elapsed / 40 * 1e3, 4),
                                        "bytes_out_per_base": 2, "one_kernel": bool(devs[0].fused), "batches_in_flight": len(devs),
                                        "bit_exact_vs_oracle": ok}
    return out


def one_shot_map_leg(args, genome, oi):
    """The call shape the reference has (lib.rs:720-761): ONE ref_seq of about the genome's length mapped against an index that is built
    for it - build, device copy and the first call all counted (a single call never reaches the bases that pay for the plan structures:
    it takes the walk + the derandomize / translate kernels), then the same call again, then with the plan structures made up front
    (kbo_index_to_device: what a caller with many sequences per index asks for)."""
    import kbo_amd
    from kbo_amd import synth
    rng = np.random.default_rng(5)
    n = min(len(genome), 5_000_000)
    ref = genome[:n].copy()
    hit = rng.random(n) < args.sub_rate
    ref[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(hit.sum()))]
    opts = kbo_amd.MapOpts(fill_gaps=False, call_variants=False, format=True)
    t0 = time.perf_counter()
    sb, _ = kbo_amd.build([genome], kbo_amd.BuildOpts(k=args.k, num_threads=min(16, usable_cores()[0])))
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    out1 = kbo_amd.map(ref, sb, None, opts)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    kbo_amd.map(ref, sb, None, opts)
    t_again = time.perf_counter() - t0
    t0 = time.perf_counter()
    sb.to_device(-1)  # (the plan structures now)
    t_plan = time.perf_counter() - t0
    t0 = time.perf_counter()
    out3 = kbo_amd.map(ref, sb, None, opts)
    t_planned = time.perf_counter() - t0
    res = {"entry_point": "kbo_build + kbo_map (fill_gaps = false, call_variants = false)", "ref_seq_bases": int(n), "index_bases": int(len(genome)),
           "build_s": round(t_build, 3), "first_map_s": round(t_first, 3), "second_map_s": round(t_again, 4),
           "plan_structures_s": round(t_plan, 3), "map_with_plan_structures_s": round(t_planned, 4),
           "end_to_end_mbp_per_s": round(n / (t_build + t_first) / 1e6, 2), "same_output_with_plan_structures": bool(out1 == out3)}
    if oi is not None:
        exp = oi.matches_batch(ref, np.array([0, n], dtype=np.uint64), 1e-7, n_threads=usable_cores()[0])
        from oracle import binding as ora
        res["bit_exact_vs_oracle"] = bool(out1 == bytes(ora.relative_to_ref(ref, exp)))
    return res


def host_to_host_leg(args, sbwt, genome):
    """The product entry point a binding calls: kbo_map_batch over pageable host buffers (H2D, kernels, D2H in a three-stage
    slab pipeline); PCIe-inclusive, never the reported value.  4 x the batch, best of 3."""
    import kbo_amd
    from kbo_amd import synth
    R = min(4 * args.reads, 4_000_000)
    concat, offsets = synth.reads(genome, R, args.read_len, args.sub_rate)
    out = np.zeros(len(concat), dtype=np.uint8)
    L = kbo_amd.lib()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        kbo_amd.check(L.kbo_map_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, R, 1e-7, 1, out.ctypes.data))
        best = min(best, time.perf_counter() - t0)
    res = {"value": round(R * args.read_len / best / 1e6, 1), "unit": "Mbp/s", "entry_point": "kbo_map_batch (format=true)",
           "reads": R, "ms": round(best * 1e3, 2), "bytes_per_base_over_pcie": 2.0,
           "note": "pageable numpy buffers in and out, 1 B/base each way; best of 4 calls (the first pays the pinned staging)"}
    # the packed entry points: 2-bit words in, 2-bit words out (kbo::matches' alphabet is M - X R), a quarter of the bytes
    from kbo_amd import batch
    words, pos, byt = batch.pack_reads(concat, offsets)
    wout = np.zeros(len(words), dtype=np.uint32)
    bestp = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        kbo_amd.check(L.kbo_matches_batch_packed(sbwt._h, words.ctypes.data, offsets.ctypes.data, R, None, None, 0, 1e-7, wout.ctypes.data))
        bestp = min(bestp, time.perf_counter() - t0)
    plain = np.zeros(len(concat), dtype=np.uint8)
    kbo_amd.check(L.kbo_matches_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, R, 1e-7, plain.ctypes.data))
    res["packed"] = {"value": round(R * args.read_len / bestp / 1e6, 1), "unit": "Mbp/s", "entry_point": "kbo_matches_batch_packed",
                     "ms": round(bestp * 1e3, 2), "bytes_per_base_over_pcie": round(2 * len(words) * 4 / (R * args.read_len), 3),
                     "equal_to_kbo_matches_batch": bool(np.array_equal(batch.unpack_matches(wout, offsets), plain))}
    return res


