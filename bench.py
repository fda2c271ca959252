#!/usr/bin/env python3
"""bench.py — kbo map query throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path - kbo::map with fill_gaps=false, call_variants=false (lib.rs:735-738, 756-757): MS ->
derandomize -> translate -> relative_to_ref - over one batch of synthetic reads already resident in HBM, through
kbo_map_batch_dev: ONE kernel for the reads (kbo_amd/csrc/map_kernels.hip) + the plain walk of the few per cent it leaves
(--two-kernels: the round-3 route, the plan-guided MS walk and the derandomize / translate kernel one after the other).
Batches are in flight on TWO PIPELINES of two streams each: a batch's kernel on its pipeline's first stream, its second pass on
the second one (kbo_map_batch_dev_tail) beside the pipeline's next kernel; consecutive steps go to the pipelines in turn and take
four resident batches of the same shape in turn (different reads, own buffers), so two kernels and two second passes share
the device at any time; every step's work - both passes - ends inside the timed region.  --one-at-a-time: a single stream
(also reported in the line: `one_batch_at_a_time`).  Default workload = BASELINE config C2: 5 Mbp iid genome, k=31 SBWT, 1 M x 150 bp forward reads with
1 % substitutions per GPU (weak scaling: every rank holds the replicated index and its own reads; no collective on the data
path).  Prints ONE JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W        N > 1 without RANK in the environment: this process builds the
                                                       index cache on the host (it never touches a GPU) and starts N
                                                       ranks through torch.distributed.run, relays rank 0's line
  --config C2 | C3 | C4                                C3 = kbo find, 100 Mbp, 10 M reads per GPU; C4 = kbo map, 250 Mbp,
                                                       100 M reads in all, sharded over the ranks (strong scaling)
The oracle (oracle/) is used after the timed region only: parity gate, CPU baseline, and the CPU model of the plan-guided
stage whose work counts price the roofline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from benchlib.common import *  # noqa: E402,F401,F403  (arguments, presets, index, ranks, the timed loops)
from benchlib.legs import *  # noqa: E402,F401,F403  (the legs behind the timed region)
from benchlib.call import main_call  # noqa: E402  (--call / --config C5)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    import benchlib.common as _common
    _common.CONDITION_MS = args.condition_ms
    if args.gpus > 1 and "RANK" not in os.environ and not args.call:
        # (this process never initialises a GPU: it builds the index cache and waits for its children)
        raise SystemExit(spawn_ranks(args, argv))
    if args.call:
        return main_call(args)
    import torch
    import torch.distributed as dist

    import kbo_amd
    from kbo_amd import batch, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # KBO_BENCH_ONE_GPU=1: functional test of the multi-process path on a single-GPU box (all ranks share
    # cuda:0 and rendezvous over gloo); numbers from such a run mean nothing
    one_gpu = os.environ.get("KBO_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # RCCL; used for barrier + max only

    L = kbo_amd.lib()
    if args.waves_per_cu:
        L.kbo_set_walk_waves_per_cu(args.waves_per_cu)
    if args.no_plan:
        L.kbo_set_plan(0, 0, 0)
    if args.depth_table:
        L.kbo_set_depth_table(args.depth_table)

    # ---- index: replicated.  Rank 0 builds it once (or finds the cache a parent / an earlier run left) and writes the
    # cache, path cover included; the other ranks load it
    threads = max(1, usable_cores()[0] // max(1, world))  # (the ranks of a node share the container's CPU quota)
    t_index = time.perf_counter()
    if world > 1:
        if rank == 0:
            genome, sbwt = build_or_load_index(args, usable_cores()[0])
        dist.barrier()
        if rank != 0:
            genome, sbwt = build_or_load_index(args, threads, may_build=not os.path.exists(cache_path(args)))
    else:
        genome, sbwt = build_or_load_index(args, threads)
    t_index = time.perf_counter() - t_index

    # ---- reads: sharded by rank.  weak scaling (C2, C3): args.reads per rank; strong (C4): args.reads in all
    n_mine, first = shard(args, rank, world)
    stream = torch.cuda.current_stream(device)
    # batches in flight (module docstring): two pipelines, on each a batch's second pass on the second stream beside the next
    # batch's kernel on the first
    piped = not args.two_kernels and not args.one_at_a_time
    n_pipes = max(1, args.pipelines) if piped else 1
    # (kbo::find has no entry point of that kind yet: its pipelines are pairs of torch streams here, kbo_find_batch_dev's tail stream)
    # (... made by the library like kbo_map_stream's own: the kernels' stream kept off 32 compute units, the tail stream plain; KBO_FIND_TAIL_CUS: experiments)
    pipes = [batch.stream_pair(device, tail_cus=int(os.environ.get("KBO_FIND_TAIL_CUS", "-1"))) for p in range(n_pipes)] if piped and args.find else None
    mstream = None
    n_slabs = (n_mine + SLAB_READS - 1) // SLAB_READS
    # (slabs of equal size: the pipelines take the slabs in turn, and with 8 M + 2 M reads - C3 - one of them had four fifths of the work)
    slab_reads = max(1, (n_mine + max(1, n_slabs) - 1) // max(1, n_slabs))
    # (several slabs per step are several batches already; fewer than two per pipeline: a second set of them)
    n_sets = (2 * n_pipes if n_slabs == 1 else 2 if n_slabs < 2 * n_pipes else 1) if piped else 1
    sets, first_slab = [], []
    for b in range(n_sets):
        slabs = []
        for s0 in range(0, n_mine, slab_reads):
            ns = min(slab_reads, n_mine - s0)
            # (the second set: the reads behind every rank's first set)
            concat, offsets = synth.reads(genome, ns, args.read_len, args.sub_rate, first_read=first + s0 + b * max(world, 1) * args.reads)
            slabs.append(batch.DeviceBatch(sbwt, concat, offsets, device=device, format=not args.find, want_ms=False))
            slabs[-1].done = None
            if s0 == 0:
                first_slab.append((concat, offsets))  # (rank 0's parity gate and CPU baseline use the first slab of either set)
        sets.append(slabs)
    slabs = sets[0]
    concat0, offsets0 = first_slab[0]
    bases = n_mine * args.read_len

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    stream0 = stream

    def one_step(step, events=None, on_tail=None):
        on_tail = piped if on_tail is None else on_tail
        for i, dev in enumerate(sets[step % n_sets]):
            if on_tail and mstream is not None:  # kbo_map_stream_*: the library's pipelines take the batches in turn
                mstream.submit(dev)
                continue
            stream, tail = pipes[(step * len(slabs) + i) % n_pipes] if on_tail else (stream0, None)
            if on_tail and dev.done is not None:
                stream.wait_event(dev.done)  # its buffers are free again behind its last second pass
            if events is not None:
                events[i][0].record(stream)
            last = tail if on_tail else stream
            if args.two_kernels:
                dev.walk(stream)
                if events is not None:
                    events[i][1].record(stream)
                dev.derand_translate(stream)
            elif args.find:  # kbo::find (lib.rs:816-820) as one call: the characters and their run lengths (kbo_find_batch_dev: the kernel
                # counts the runs of the reads it finishes, the records are one more pass over the characters)
                dev.run_find(0, stream, tail_stream=tail if on_tail else None)
                if events is not None:
                    for e in events[i][1:]:
                        e.record(last)
            else:  # kbo_map_batch_dev[_tail]: one kernel for the reads + the plain walk of the reads it leaves
                dev.run(stream, tail_stream=tail if on_tail else None)
                if events is not None:
                    events[i][1].record(last)
            if events is not None and args.two_kernels:
                events[i][2].record(last)
            if args.find and args.two_kernels:  # (the round-3 route: run lengths as a call of their own)
                dev.run_lengths(0, last)
                if events is not None:
                    events[i][3].record(last)
            if on_tail:
                if dev.done is None:
                    dev.done = torch.cuda.Event()
                dev.done.record(tail)

    for b in range(n_sets):  # every resident batch once before anything else: which route it takes (and the lazy parts of the copy)
        one_step(b, None, on_tail=False)
    sync_all()
    one_kernel = (not args.two_kernels) and all(d.fused for sl in sets for d in sl)
    piped = piped and one_kernel  # (the two-kernel route has no second pass to set aside)
    if piped and not args.find:
        every = [d for sl in sets for d in sl]
        mstream = batch.MapStream(sbwt, max(d.n_seqs for d in every), max(d.total for d in every), max(d.max_len for d in every), pipelines=n_pipes)
    # everything the timed region needs is made BEFORE the warm-up steps, so that the timed steps follow them at once: a collector
    # run and a few hundred event objects between the two are tens of milliseconds of an idle device, and the first kernels behind
    # an idle device run slower than the rest (its clocks) - which 20 timed steps feel
    ev = [[[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in slabs] for _ in range(args.steps)]
    import ctypes as _C
    import gc
    gc.collect()
    gc.disable()  # (the timed region is tens of milliseconds of enqueueing: no collector pause inside it)
    # (event records per call: the dominant kernel's own duration, live; the library makes the events of so many calls now)
    L.kbo_set_stage_timing((args.warmup + args.steps + 4) * len(slabs) + 1 if one_kernel else 0)
    # the device has idled through the set-up above: the first ~15 ms of load behind that run 5 - 10 % slower than the same work later
    # (tools/ramp_timeline.py: 20 batches behind 5 / 20 / 100 warm ones 886 / 858 / 947 Gbp/s), which --steps 20 --warmup 5 would measure
    # instead of the path.  Untimed steps of the same workload for --condition-ms first; the line says how many
    conditioned = 0
    if args.condition_ms > 0:
        t_cond = time.perf_counter()
        for j in range(4):
            one_step(j)
        torch.cuda.synchronize(device)
        per_step = max(1e-6, (time.perf_counter() - t_cond) / 4)
        conditioned = 4 + min(4000, int(args.condition_ms * 1e-3 / per_step) + 1)  # (enqueued back to back: the warm-up steps follow with no gap)
        for j in range(4, conditioned):
            one_step(j)
    for w in range(args.warmup):
        one_step(conditioned + w)
    sync_all()
    L.kbo_stage_timing_read(None, None, None)  # (the untimed steps' records: forgotten)
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(conditioned + args.warmup + s, ev[s])
    sync_all()
    elapsed = time.perf_counter() - t0
    gc.enable()
    L.kbo_set_stage_timing(0)
    k_sum, r_sum, n_calls = _C.c_double(0), _C.c_double(0), _C.c_int(0)
    L.kbo_stage_timing_read(_C.byref(k_sum), _C.byref(r_sum), _C.byref(n_calls))
    map_kernel_ms = k_sum.value / args.steps if one_kernel and n_calls.value else None  # per step (all slabs)
    map_redo_ms = r_sum.value / args.steps if one_kernel and n_calls.value else None
    # (a call's span: from events around it - or, through kbo_map_stream_*, whose streams are the library's, the two intervals it timed itself)
    walk_ms = (map_kernel_ms + map_redo_ms) if mstream is not None and map_kernel_ms is not None else \
        float(np.mean([sum(e[0].elapsed_time(e[1]) for e in step) for step in ev]))
    dt_ms = float(np.mean([sum(e[1].elapsed_time(e[2]) for e in step) for step in ev])) if args.two_kernels else 0.0
    rle_ms = float(np.mean([sum(e[2].elapsed_time(e[3]) for e in step) for step in ev])) if args.find and args.two_kernels else None  # (else: inside the call)
    # what the timed steps left behind (formatted unless --find): the first slab of either set, for rank 0's parity gate
    timed_chars_sets = [sl[0].chars[:sl[0].total].cpu().numpy() for sl in sets] if rank == 0 else None
    if mstream is not None:  # (its streams go back to the runtime: the legs below make their own, and streams beyond the hardware queues share them)
        mstream.close()
        mstream = None
    serial = None
    if piped and rank == 0 and world == 1:  # the same steps on one stream, for the record
        n_ser = max(1, min(args.steps, 10))
        condition(lambda n: [one_step(j, None, on_tail=False) for j in range(n)], 2, torch, device)  # (the parity copies above were host work)
        sync_all()
        L.kbo_set_stage_timing(1 if one_kernel else 0)
        t1 = time.perf_counter()
        for s in range(n_ser):
            one_step(s, None, on_tail=False)
        sync_all()
        t1 = time.perf_counter() - t1
        L.kbo_set_stage_timing(0)
        ks, rs, nc = _C.c_double(0), _C.c_double(0), _C.c_int(0)
        L.kbo_stage_timing_read(_C.byref(ks), _C.byref(rs), _C.byref(nc))
        serial = {"value": round(bases * n_ser / t1 / 1e6, 1), "unit": "Mbp/s", "ms_per_step": round(t1 / n_ser * 1e3, 4), "steps": n_ser,
                  "map_reads_kernel_ms": round(ks.value / n_ser, 4) if one_kernel and nc.value else None,
                  "second_pass_ms": round(rs.value / n_ser, 4) if one_kernel and nc.value else None,
                  "note": "kbo_map_batch_dev on one stream: every batch's second pass before the next batch's kernel"}

    t = torch.tensor([elapsed], dtype=torch.float64, device=device if not one_gpu or world == 1 else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # per-rank stage times (skew between ranks shows here)
    per_rank_t = torch.tensor([map_kernel_ms if map_kernel_ms is not None else walk_ms, dt_ms], dtype=torch.float64, device=t.device)
    if world > 1:
        gathered = [torch.zeros_like(per_rank_t) for _ in range(world)]
        dist.all_gather(gathered, per_rank_t)
        walk_all = [float(g[0].item()) for g in gathered]
    else:
        walk_all = [float(per_rank_t[0].item())]
    total_bases = args.reads * args.read_len if args.scaling == "strong" else world * bases

    result = None
    if rank == 0:
        dev = slabs[0]
        # the stage's own work counters: one more launch over the first slab with the counting instantiations of the
        # kernels (instrumentation, off inside the timed region: about 1 % of the stage's time)
        timed_chars = timed_chars_sets[0]
        L.kbo_set_plan_stats(1)
        if one_kernel:
            dev.run(stream)
            torch.cuda.synchronize(device)
            map_stats = dev.plan_stats(stream)
        dev.walk(stream)
        torch.cuda.synchronize(device)
        stats = dev.plan_stats(stream)
        L.kbo_set_plan_stats(0)
        gpu_d = dev.ms.cpu().numpy()
        cpu = b_ref = exact = ops = b_plan = model = None
        sens = h2h = ms_var = one_shot = None
        if world == 1 and not args.no_cpu_baseline and not args.no_extras and (args.extras or not args.custom):
            # (first of the legs behind the timed region: its pinned staging buffers are made by its first call, and behind the
            # oracle's and the variants' gigabytes of host allocations they come out of scattered pages - 21 instead of 40 Gbp/s)
            h2h = host_to_host_leg(args, sbwt, genome)
        planned = (not args.no_plan) and sbwt.device_plan_bytes() > 0
        if not args.no_cpu_baseline:
            from oracle import binding as ora
            rows, Carr, lcs = sbwt.export_parts()
            # (the oracle adopts the product-built index: its own row-sorting builder needs minutes and > 30 B/base beyond
            # ~20 Mbp; builder equality is tests/test_builder_vs_oracle.py)
            oi = ora.Index.from_parts(args.k, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
            # parity gate (rank 0's first slab) + CPU baseline on the unformatted characters, at every world size
            fmt = dev.format
            dev.format = False
            dev.derand_translate(stream)
            torch.cuda.synchronize(device)
            cpu, b_ref, exact, ops = cpu_baseline_leg(args, oi, concat0, offsets0, gpu_d, dev.chars.cpu().numpy())
            dev.format = fmt
            # EVERY read of the slab: the characters the timed steps wrote (kbo::map's output) and the MS values, against the oracle
            exp_chars, exp_d = oi.matches_batch(concat0, offsets0, 1e-7, n_threads=usable_cores()[0], want_d=True)
            exp_out = np.frombuffer(ora.relative_to_ref(concat0, exp_chars), dtype=np.uint8) if fmt else exp_chars
            exact = bool(exact and np.array_equal(timed_chars, exp_out) and np.array_equal(gpu_d[:dev.total], exp_d))
            for b in range(1, n_sets):  # ... and of the other batch in flight
                cb, ob = first_slab[b]
                exp_chars = oi.matches_batch(cb, ob, 1e-7, n_threads=usable_cores()[0])
                exp_out = np.frombuffer(ora.relative_to_ref(cb, exp_chars), dtype=np.uint8) if fmt else exp_chars
                exact = bool(exact and np.array_equal(timed_chars_sets[b], exp_out))
            del exp_chars, exp_d, exp_out
            if planned and not stats["gave_up"] and not one_kernel:
                b_plan, model, _ = stage_model_leg(args, sbwt, oi, concat0, offsets0, gpu_d)
                exact = bool(exact and model["ms_equal_to_gpu"])
            if world == 1 and not args.no_extras and (args.extras or not args.custom):
                sens = sensitivity_leg(args, genome, sbwt, oi, torch, device, stream, n_pipes if piped else None)
                ms_var = ms_leg(args, sbwt, oi, concat0, offsets0, torch, device, stream, n_pipes if piped else None)
                one_shot = one_shot_map_leg(args, genome, oi) if args.genome <= 20_000_000 else None
        # fabric-side traffic and L2 misses of the A1 stage, from the committed rocprofv3 passes of this exact
        # workload and walk mode (PMC passes cannot run inside the timed region: separate runs, tools/profile_bench.sh)
        wl_key = f"{args.genome}x{n_mine}x{args.read_len}x{args.sub_rate:g}:{('map' if one_kernel else 'table' if sbwt.depth_table_order() > 0 else 'plan') if planned else 'plain'}"
        sha = build_sha16()
        traffic = tsrc = misses = walk_misses = plan_misses = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                entry = json.load(open(tpath)).get("workloads", {}).get(wl_key)
                if entry and one_kernel and entry.get("build_sha16") != sha:
                    tsrc = "none: %s was taken of another build (%s, this one is %s) - tools/profile_bench.sh renews it" % (
                        entry.get("source"), entry.get("build_sha16"), sha)
                    entry = None
                if entry:
                    traffic, misses, tsrc = entry.get("a1_bytes_per_launch"), entry.get("a1_tcc_miss_per_launch"), entry.get("source")
                    kern = entry.get("kernels", {})
                    walk_misses = (kern.get("ms_walk_guided_kernel") or kern.get("ms_walk_recovery_kernel") or {}).get("tcc_miss")
                    plan_misses = (kern.get("plan_kernel") or {}).get("tcc_miss")
            except Exception:
                pass
        lay = sbwt.device_layout()
        rank_b, lcs_b = sbwt.device_bytes()
        dto = sbwt.depth_table_order()
        dtab_b = 0 if dto == 0 else (4 ** (dto + 1) if dto >= 4 else 4 ** dto)  # (grouped from 4 bases on: DESIGN.md section 4.2)
        seed_d = int(lay["seed_depth"]) if planned else 0  # (from the library: kbo_index_device_layout - what the copy really holds)
        seed_b = 8 * 4 ** seed_d if seed_d else 0
        pair_b, plan_b = sbwt.device_pair_bytes(), sbwt.device_plan_bytes()
        std = (args.genome, args.reads, args.read_len, args.sub_rate, args.k) in tuple((g, r, 150, 0.01, 31) for g, r, _, _ in PRESETS.values())
        label = (args.config if std else "custom") + ": " + \
            ("kbo find (max_gap_len=0; run lengths on the device)" if args.find else
             "kbo map (fill_gaps=false, call_variants=false, format=true)")
        table = planned and sbwt.depth_table_order() > 0
        a1_kernels = ("plan_kernel (with the depth-table look-ups; dtab_resolve_kernel for items that cannot be staged) + redo_collect + ms_walk_kernel (reads the table could not resolve)" if table
                      else "plan_kernel + plan_count/scan/emit + ms_walk_guided_kernel (ms_walk_recovery_kernel from 24 Mi rows on) + redo_collect + ms_walk_kernel (flagged reads)"
                      if planned else "ms_walk_kernel")
        walk_s = walk_ms * 1e-3
        # what the stage is priced by: its OWN compulsory bytes (B_plan, counted by the model on the timed reads) when it
        # planned, the reference algorithm's bytes when it walked plainly
        b_alg = b_plan if b_plan is not None else b_ref
        achieved = b_alg * bases / walk_s / 1e9 if b_alg is not None else None
        ref_achieved = b_ref * bases / walk_s / 1e9 if b_ref is not None else None
        roofline = {
            "bound": "hbm",
            "bound_detail": ("independent random byte gathers from the depth table (one fill each) + plan_kernel's streams and compare loop "
                             "(DESIGN.md section 4.2)" if table else
                             "L2-miss line fills by their rate and latency: integer gather work, 16 bytes used per 128-byte fill; the guided walk "
                             "keeps 8-12 waves per CU so that the lines of the lanes in flight stay in L2 (DESIGN.md sections 4.2, 6)"),
            "achieved": round(achieved, 1) if achieved is not None else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4) if achieved is not None else None,
            "frac_meaning": ("compulsory bytes of the plan-guided stage (B_plan: streams, records, " +
                             ("one table byte per look-up and the query window of every mismatch" if table else "two 16-byte loads per walk iteration") +
                             "; counted on the timed reads by oracle/plan_model.c, whose counts equal the kernels' own counters) / A1 stage time / 8 TB/s"
                             if b_plan is not None else "reference-algorithm bytes (SURVEY.md 8(d)) / A1 stage time / 8 TB/s"),
            "algorithmic_bytes_per_base": round(b_alg, 3) if b_alg is not None else None,
            "units_per_launch": bases, "kernel": "A1 stage = " + a1_kernels, "kernel_ms": round(walk_ms, 4),
            "kernel_ms_per_rank": {"min": round(min(walk_all), 4), "max": round(max(walk_all), 4)},
            "traffic": int(traffic) if traffic else None, "traffic_source": tsrc,
            "traffic_frac": round(traffic / walk_s / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
            "wasted_traffic": round(traffic / (b_alg * bases), 3) if traffic and b_alg else None,
            "l2_miss_per_launch": int(misses) if misses else None,
            "fill_rate_frac": round(misses / walk_s / FILL_CEILING_PER_S, 4) if misses else None,
            "fill_rate_ceiling": {"fills_per_s": FILL_CEILING_PER_S, "source": FILL_CEILING_SOURCE},
            "fills_min_per_unit": model["per_unit"]["distinct_lines_beyond_l2"] if model and "per_unit" in model else None,
            "fills_min_per_read": model.get("fills_min_per_read") if model else None,
            "l2_miss_per_read": round(misses / (bases / args.read_len), 2) if misses else None,
            # measured L2 misses per unit: of the whole stage, and of the walk kernel alone (to set against fills_min_per_unit:
            # the difference is the rank-block look-ups that miss although the blocks would fit the L2)
            "l2_miss_per_unit": round(misses / max(1, stats["units"] * bases / dev.total), 2) if misses and planned and not table else None,
            "walk_kernel_l2_miss_per_unit": round(walk_misses / max(1, stats["units"] * bases / dev.total), 2) if walk_misses and planned and not table else None,
            "plan_kernel_l2_miss_per_read": round(plan_misses / (bases / args.read_len), 2) if plan_misses and table else None,
            "frac_reference_algorithm": round(ref_achieved / HBM_PEAK_GBPS, 4) if ref_achieved is not None else None,
            "reference_algorithm_bytes_per_base": round(b_ref, 2) if b_ref is not None else None,
            "cross_check_whole_step_gbps": round(b_alg * bases / (elapsed / args.steps) / 1e9, 1) if b_alg is not None else None,
            "stage_model": model,
            "stage_counters_gpu_first_slab": stats if planned else None,
        }
        if one_kernel:
            # the dominant kernel priced by its OWN compulsory bytes, counted by the kernel itself on the first slab (kbo_set_plan_stats:
            # one extra launch behind the timed region): what it must read and write however well it is written
            c = map_stats
            seeded = dev.n_seqs - c["items_noplan"]
            by = {"query_bytes_in": dev.total, "characters_out": dev.total, "offsets_and_flags": 9 * dev.n_seqs,
                  "seed_positions": 4 * c["seed_lookups"], "text_2bit_and_marks": 96 * seeded, "depth_table_bytes": c["tab_lookups"],
                  "filter_words": 4 * c["seed_extensions"]}  # (seed_extensions: this kernel counts its filter look-ups there)
            b_map = sum(by.values()) / dev.total
            # the kernel's duration, launch by launch (HIP events around it).  With two pipelines two launches share the device
            # for the whole of their durations: the device's time per launch is then the timed region / launches, not a launch's
            # own duration - the bytes are priced by that (never less than duration / pipelines), the duration is printed beside it
            k_launch_s = map_kernel_ms * 1e-3
            shared = piped and n_pipes > 1
            k_s = max(elapsed / args.steps, k_launch_s / n_pipes) if shared else k_launch_s
            achieved = b_map * bases / k_s / 1e9
            lines_min = (2 * dev.total / 128 + c["seed_lookups"] + c["tab_lookups"]) / dev.n_seqs  # streams + one line per table access (the filter's 2 MB stay in L2)
            roofline = {
                "bound": "hbm",
                "bound_detail": "integer gather work, no MFMA: per read one seed-position look-up, three depth-table bytes per mismatch (each a line "
                                "of its own; four in five settled by a 2 MB filter in L2 before they reach the table), the 2-bit text on the diagonal "
                                "(L2 / Infinity Cache) and 2 B per base of streams.  Alone the kernel is not bound by its fills (with or without the "
                                "filter: the same time), nor by its instructions: the wave's own chain of dependent loads is what is left (DESIGN.md 4.1)"
                                + ("; the timed kernels ran beside the other batches' kernels and second passes (batches_in_flight)" if piped else ""),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "frac_meaning": "compulsory bytes of map_reads_kernel (counted by the kernel on the timed reads: bytes_by_part) x bases per step / " +
                                ("the device's time per launch / 8 TB/s.  %d launches share the device at any time (one per pipeline), each for the "
                                 "whole of its duration (kernel_ms: HIP events around every launch of the timed region, what rocprofv3 shows per "
                                 "dispatch), so the device's time per launch is the timed region / launches = ms_per_step, second passes included "
                                 "(never taken below kernel_ms / pipelines).  per_launch_duration prices the same bytes by kernel_ms as if the "
                                 "launch had the device to itself; alone is the kernel with nothing beside it (one_batch_at_a_time)" % n_pipes
                                 if shared else "its own duration (HIP events around it in every timed step) / 8 TB/s"),
                "launches_sharing_the_device": n_pipes if shared else 1,
                "device_ms_per_launch": round(k_s * 1e3, 4),
                "per_launch_duration": {"achieved": round(b_map * bases / k_launch_s / 1e9, 1), "frac": round(b_map * bases / k_launch_s / 1e9 / HBM_PEAK_GBPS, 4)},
                "alone": ({"kernel_ms": serial["map_reads_kernel_ms"], "achieved": round(b_map * bases / (serial["map_reads_kernel_ms"] * 1e-3) / 1e9, 1),
                           "frac": round(b_map * bases / (serial["map_reads_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
                          if serial and serial.get("map_reads_kernel_ms") else None),
                "algorithmic_bytes_per_base": round(b_map, 3), "bytes_by_part_first_slab": by, "units_per_launch": bases,
                "kernel": "map_reads_kernel (kbo_amd/csrc/map_kernels.hip): MS -> derandomize -> translate -> relative_to_ref of every read it can finish",
                "kernel_ms": round(map_kernel_ms, 4), "kernel_ms_per_rank": {"min": round(min(walk_all), 4), "max": round(max(walk_all), 4)},
                "redo_pass_ms": round(map_redo_ms, 4),
                "redo_pass": "finish_reads_kernel over the %.2f %% of the reads the kernel leaves: a wave a read, its lanes walk pieces of it "
                             "behind k - 1 warm-up bases (a chain of about k + 10 dependent look-ups), then derandomize + translate and the "
                             "characters" % (100.0 * c["tab_unresolved"] / dev.n_seqs),
                "traffic": int(traffic) if traffic else None, "traffic_source": tsrc,
                "traffic_frac": round(traffic / k_s / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
                "wasted_traffic": round(traffic / (b_map * bases), 3) if traffic else None,
                "l2_miss_per_launch": int(misses) if misses else None,
                "fill_rate_frac": round(misses / k_s / FILL_CEILING_PER_S, 4) if misses else None,
                "fill_rate_ceiling": {"fills_per_s": FILL_CEILING_PER_S, "source": FILL_CEILING_SOURCE},
                "fills_min_per_read": round(lines_min, 2), "l2_miss_per_read": round(misses / (bases / args.read_len), 2) if misses else None,
                "frac_reference_algorithm": round(b_ref * bases / k_s / 1e9 / HBM_PEAK_GBPS, 4) if b_ref is not None else None,
                "reference_algorithm_bytes_per_base": round(b_ref, 2) if b_ref is not None else None,
                "cross_check_whole_step_gbps": round(b_map * bases / (elapsed / args.steps) / 1e9, 1),
                "stage_counters_gpu_first_slab": c, "build_sha16": sha,
            }
        result = {
            "metric": f"query Mbp/sec for kbo {'find' if args.find else 'map'}, k={args.k}, {args.genome / 1e6:g} Mbp SBWT; bit-exact MS vs CPU",
            "value": round(total_bases * args.steps / elapsed / 1e6, 1),
            "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "untimed_before_warmup": {"steps": conditioned, "condition_ms": args.condition_ms,
                                      "why": "the same steps, untimed, in front of the warm-up ones: the device leaves the idle set-up phase "
                                             "with low clocks, and 5 warm-up steps are under 1 ms of load (--condition-ms 0: none)"},
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{label}, {args.genome / 1e6:g} Mbp iid genome SBWT k={args.k}, "
                                   + (f"{args.reads} x {args.read_len} bp reads in all, {n_mine} per GPU, " if args.scaling == "strong"
                                      else f"{args.reads} x {args.read_len} bp reads per GPU, ")
                                   + f"{args.sub_rate * 100:g}% substitutions",
                       "index_n_sets": sbwt.n_sets(), "threshold": dev.threshold,
                       "timed_output": ("format::run_lengths_gapped records + the characters" if args.find else "kbo::map's characters, one byte a base") +
                                       "; the matching statistics of every read are compared with the oracle's behind the timed region, and the entry "
                                       "points that RETURN them are timed in ms_variant",
                       "walk": ("kbo_map_batch_dev: one kernel per batch of reads (path cover as 2-bit text, seed positions, depth table of %d bases)" % sbwt.depth_table_order()
                                if one_kernel else "plan-guided (path cover + depth table of %d bases)" % sbwt.depth_table_order() if table else
                                "plan-guided (path cover + guided walk)" if planned else "plain"),
                       # (from the library: kbo_index_device_layout - what the copy really holds, and what making it cost)
                       "index_device_bytes": {kk: v for kk, v in lay.items() if kk.endswith("_bytes")} | {
                           "total": sum(v for kk, v in lay.items() if kk.endswith("_bytes")),
                           "per_row_without_tables": round(sum(lay[kk] for kk in ("rank_bytes", "entry_bytes", "pair_bytes", "cover_bytes", "lines_bytes")) / sbwt.n_sets(), 2),
                           "depth_table_order": lay["dtab_order"], "depth_table_layout": "grouped" if lay["dtab_grouped"] else "plain",
                           "seed_table_depth": lay["seed_depth"], "entries_64bit": lay["entries_64bit"],
                           "note": "the seed table(s) and the depth table are sized by log4(rows), not by the index"},
                       "setup_seconds": {kk[:-8]: round(v, 3) for kk, v in lay.items() if kk.endswith("_seconds")} | {
                           "total": round(sum(v for kk, v in lay.items() if kk.endswith("_seconds")), 3),
                           "note": "this rank's device copy: host layout of rank blocks / entries, uploads, path cover (0 when the index "
                                   "file carried it), recovery lines, seed table(s), depth table; index build or load is index_seconds_rank0"},
                       "resident_slabs_per_gpu": len(slabs), "index_seconds_rank0": round(t_index, 2),
                       "batches_in_flight": ("%d on %d pipeline(s) of two streams (kbo_map_stream_*: the library's own; kbo::find: kbo_find_batch_dev's tail stream): consecutive launches go to the pipelines in turn and take %s in turn; on a "
                                             "pipeline a batch's second pass runs on the second stream beside the next "
                                             "batch's kernel on the first; all of every step's work ends inside the timed region"
                                             % (2 * n_pipes, n_pipes, ("%d resident batches of this shape (different reads)" % n_sets) if n_sets > 1 else "the step's slabs")) if piped else 1,
                       "parallelism": f"index replicated x{world}, reads sharded, no collective"},
            "roofline": roofline,
            "kernels_ms": ({"map_reads_kernel": round(map_kernel_ms, 4), "redo_pass": round(map_redo_ms, 4),
                            "kbo_map_batch_dev": round(walk_ms, 4)} if one_kernel else
                           {"a1_stage": round(walk_ms, 4), "derand_translate": round(dt_ms, 4)}) |
                          ({"run_lengths": round(rle_ms, 4)} if rle_ms is not None else {}),
            "one_batch_at_a_time": serial,
            "cpu_baseline": cpu,
            "bit_exact_vs_oracle": exact,
            "reference_ops_per_base": ops,
            "sensitivity": sens,
            "ms_variant": ms_var,
            "one_shot_map": one_shot,
            "host_to_host": h2h,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
