#!/usr/bin/env python3
"""bench.py — kbo map query throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (A1 walk kernel + fused A5/A6 derandomize/translate
kernel, i.e. kbo::map with fill_gaps=false, call_variants=false, lib.rs:735-738) over one
batch of synthetic reads already resident in HBM.  Workload = BASELINE config C2:
5 Mbp iid genome, k=31 SBWT, 1 M x 150 bp forward reads with 1 % substitutions per GPU
(weak scaling: every rank holds the replicated index and its own reads; no collective on
the data path).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["C2", "C3"], default="C2",
                    help="BASELINE.json workload: C2 = kbo map, 5 Mbp index, 1 M x 150 bp reads (the metric config); "
                         "C3 = kbo find, 100 Mbp index, 10 M x 150 bp reads (SURVEY.md 8(d)'s designated roofline run)")
    ap.add_argument("--genome", type=int, default=None)
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub-rate", type=float, default=0.01)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall-clock budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--find", action="store_true",
                    help="time kbo::find instead of kbo::map: the step ends with format::run_lengths on the device")
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--call", action="store_true",
                    help="time the first pass of kbo call (C5 shape, scaled): MS walk with intervals + the breakpoint scan "
                         "on the device over 10 kbp reads (defaults: --genome 100000000 --reads 10000 --read-len 10000)")
    ap.add_argument("--no-plan", action="store_true", help="plain walk kernel only (no path cover, no plan-guided walk)")
    args = ap.parse_args()
    if args.call:
        args.genome = args.genome if args.genome is not None else 100_000_000
        args.reads = args.reads if args.reads is not None else 10_000
        args.read_len = args.read_len if args.read_len != 150 else 10_000
        args.custom = True
        return args
    preset = {"C2": (5_000_000, 1_000_000, False), "C3": (100_000_000, 10_000_000, True)}[args.config]
    args.custom = args.genome is not None or args.reads is not None
    args.genome = args.genome if args.genome is not None else preset[0]
    args.reads = args.reads if args.reads is not None else preset[1]
    args.find = args.find or (preset[2] and not args.custom)
    return args


def usable_cores():
    """Threads worth starting: the CPUs this process may run on, capped by the container's CFS quota (the GPU boxes
    show 256 hardware threads but grant 16 CPUs' worth of time; more threads than that only get throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} schedulable CPUs"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period + 0.5))
                if q < n:
                    note = f"cgroup CPU quota {q} of {n} schedulable CPUs"
                    n = q
            break
        except Exception:
            continue
    return n, note


def cpu_baseline_leg(args, genome, concat, offsets, gpu_d, gpu_chars, sbwt=None):
    """Times the oracle (C restatement of the reference algorithm, sbwt-like layout) on a
    bounded sample of the same reads with all host cores, checks the GPU output against
    it, and returns (cpu_baseline dict, B_alg bytes/base, bit_exact)."""
    from oracle import binding as ora
    cores, cores_note = usable_cores()
    if len(genome) <= 20_000_000 or sbwt is None:
        oi = ora.Index.build([genome.tobytes()], k=args.k)
    else:
        # the oracle's own row-sorting builder needs minutes and > 30 B/base beyond ~20 Mbp; for the
        # large configs it adopts the product-built index (builder equality is a separate CPU test)
        rows, Carr, lcs = sbwt.export_parts()
        oi = ora.Index.from_parts(args.k, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
    L = args.read_len
    # calibration slice (also warms the index), then a sample sized to the time budget: as many of the reads as fit,
    # walked `passes` times by a pinned thread pool after an untimed warm-up pass (oracle/kbo_oracle.c
    # ora_matches_batch_timed: outputs allocated and touched beforehand, reads handed out dynamically)
    n0 = min(args.reads, 20_000)
    _, _, dt0 = oi.matches_batch_timed(concat[:n0 * L], offsets[:n0 + 1], 1e-7, n_threads=cores, passes=1)
    dt0 = max(dt0, 1e-4)
    n1 = int(min(args.reads, max(n0, n0 * args.cpu_seconds / dt0)))
    passes = int(max(1, min(50, args.cpu_seconds / max(dt0 * n1 / n0, 1e-3))))
    chars, d, sec = oi.matches_batch_timed(concat[:n1 * L], offsets[:n1 + 1], 1e-7, n_threads=cores, passes=passes)
    dt = sec / passes
    # operation counts of the reference algorithm (separate, untimed, counted run)
    ctr = ora.Counters()
    nc = min(n1, 50_000)
    oi.matches_batch(concat[:nc * L], offsets[:nc + 1], 1e-7, n_threads=cores, counters=ctr)
    c = ctr.as_dict()
    b_alg = (64.0 * c["rank_blocks"] + 1.0 * c["lcs_reads"]) / c["bases"] + 2.0
    exact = bool(np.array_equal(d, gpu_d[:n1 * L]) and np.array_equal(chars, gpu_chars[:n1 * L]))
    # single-thread rate of the same restatement (SURVEY.md section 8(d) asks for both), same driver, ~1 s sample
    ns = int(max(1, min(n1, 2_000_000 // L)))
    _, _, sec1 = oi.matches_batch_timed(concat[:ns * L], offsets[:ns + 1], 1e-7, n_threads=1, passes=1, want_d=False)
    single = ns * L / max(sec1, 1e-6) / 1e6
    allcore = n1 * L / dt / 1e6
    base = {"value": round(allcore, 3), "unit": "Mbp/s", "cores": cores, "kind": "port",
            "single_thread_value": round(single, 3),
            "scaling_efficiency": round(allcore / max(single * cores, 1e-9), 3), "cores_note": cores_note,
            "sample": f"first {n1} of the {args.reads} reads ({n1 * L / 1e6:.1f} Mbp) x {passes} timed passes after a warm-up "
                      f"pass, oracle/kbo_oracle.c ora_matches_batch_timed on a pool of {cores} pinned threads, "
                      f"{sec:.1f} s wall ({sec * cores:.0f} core-seconds)"}
    ops = {k: round(v / c["bases"], 4) for k, v in c.items() if k != "bases"}
    return base, b_alg, exact, ops


def main_call(args):
    """kbo call, first pass (variant_calling.rs:266-273) over a batch of long reads resident in HBM: A1 with intervals,
    then the breakpoint scan; what leaves the device is one 16-byte record per site.  Parity: the sites of the first
    reads against a host scan of the oracle's MS."""
    import torch
    import kbo_amd
    from kbo_amd import batch, derandomize, synth
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    genome = synth.genome(args.genome)
    cores, _ = usable_cores()
    sbwt, _ = kbo_amd.build([genome], kbo_amd.BuildOpts(k=args.k, num_threads=min(16, cores)))
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
    # {offset of i, offset of j, row, 0} -> {read, i, j, row}
    recs = np.stack([raw[:, 0] // args.read_len, raw[:, 0] % args.read_len, raw[:, 1] % args.read_len, raw[:, 2]], axis=1) if n_sites else raw
    # parity: sites of the first reads vs a host scan (variant_calling.rs:268-273) of the oracle's MS
    exact = None
    if not args.no_cpu_baseline:
        from oracle import binding as ora
        rows, Carr, lcs = sbwt.export_parts()
        oi = ora.Index.from_parts(args.k, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
        n_chk = min(args.reads, 40)
        want = set()
        for r in range(n_chk):
            q = concat[r * args.read_len:(r + 1) * args.read_len].tobytes()
            d, lo, hi = oi.matching_statistics(q)
            for i in range(1, len(q)):
                if d[i] < d[i - 1] and d[i - 1] >= thr and d[i] < thr:
                    for j in range(i + 1, min(i + args.k + 1, len(q))):
                        if d[j] >= thr and hi[j] - lo[j] == 1:
                            want.add((r, i, j, int(lo[j])))
                            break
        got = {tuple(int(v) for v in x) for x in recs if x[0] < n_chk}
        exact = got == want and fits
    bases = dev.total
    print(json.dumps({
        "metric": f"query Mbp/sec for kbo call first pass (MS walk whose lanes run the breakpoint scan; sites only leave the device), k={args.k}, "
                  f"{args.genome / 1e6:g} Mbp SBWT",
        "value": round(bases * args.steps / elapsed / 1e6, 1), "unit": "Mbp/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"C5 shape, scaled: kbo call first pass, {args.genome / 1e6:g} Mbp iid genome SBWT k={args.k}, "
                               f"{args.reads} x {args.read_len} bp reads, {args.sub_rate * 100:g}% substitutions",
                   "threshold": thr, "sites_per_step": n_sites, "bytes_leaving_the_device_per_base": round(16 * n_sites / bases, 4)},
        "kernels_ms": {"ms_walk_call_mode": round(walk_ms, 4)},
        "bit_exact_vs_oracle": exact}), flush=True)


def main():
    args = parse()
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

    if args.waves_per_cu:
        kbo_amd.lib().kbo_set_walk_waves_per_cu(args.waves_per_cu)
    if args.no_plan:
        kbo_amd.lib().kbo_set_plan(0, 0, 0)

    # ---- inputs (deterministic); index replicated, reads sharded by rank
    genome = synth.genome(args.genome)
    threads = max(1, usable_cores()[0] // max(1, world))  # (the ranks of a node share the container's CPU quota)
    sbwt, _ = kbo_amd.build([genome], kbo_amd.BuildOpts(k=args.k, num_threads=min(16, threads)))
    concat, offsets = synth.reads(genome, args.reads, args.read_len, args.sub_rate,
                                  first_read=rank * args.reads)
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=device, format=not args.find)
    bases = args.reads * args.read_len
    stream = torch.cuda.current_stream(device)

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        dev.run(stream)
        if args.find:
            dev.run_lengths(0, stream)
    sync_all()

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record(stream)
        dev.walk(stream)
        ev[s][1].record(stream)
        dev.derand_translate(stream)
        ev[s][2].record(stream)
        if args.find:  # kbo::find (lib.rs:816-820): run lengths of the characters, still on the device
            dev.run_lengths(0, stream)
        ev[s][3].record(stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    walk_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    dt_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    rle_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev])) if args.find else None

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # per-rank stage times (skew between ranks shows here)
    per_rank = torch.tensor([walk_ms, dt_ms], dtype=torch.float64, device=device)
    if world > 1:
        gathered = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(gathered, per_rank)
        walk_all = [float(g[0].item()) for g in gathered]
    else:
        walk_all = [walk_ms]

    result = None
    if rank == 0:
        gpu_d = dev.ms.cpu().numpy()
        cpu, b_alg, exact, ops = None, None, None, None
        if not args.no_cpu_baseline:
            # parity gate (rank 0's shard) + CPU baseline on the unformatted characters, at every world size
            fmt = dev.format
            dev.format = False
            dev.derand_translate(stream)
            torch.cuda.synchronize(device)
            cpu, b_alg, exact, ops = cpu_baseline_leg(args, genome, concat, offsets, gpu_d,
                                                      dev.chars.cpu().numpy(), sbwt)
            dev.format = fmt
        achieved = b_alg * bases / (walk_ms * 1e-3) / 1e9 if b_alg is not None else None
        # fabric-side traffic and L2 misses of the A1 stage, from the committed rocprofv3 passes of this exact
        # workload and walk mode (PMC passes cannot run inside the timed region: separate runs, tools/profile_bench.sh)
        planned = (not args.no_plan) and sbwt.device_plan_bytes() > 0
        wl_key = f"{args.genome}x{args.reads}x{args.read_len}x{args.sub_rate:g}:{'plan' if planned else 'plain'}"
        traffic = tsrc = misses = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                entry = json.load(open(tpath)).get("workloads", {}).get(wl_key)
                if entry:
                    traffic, misses, tsrc = entry.get("a1_bytes_per_launch"), entry.get("a1_tcc_miss_per_launch"), entry.get("source")
            except Exception:
                pass
        rank_b, lcs_b = sbwt.device_bytes()
        pair_b, plan_b = sbwt.device_pair_bytes(), sbwt.device_plan_bytes()
        resident = rank_b + lcs_b + pair_b < 200e6
        std = (args.genome, args.reads, args.read_len, args.sub_rate, args.k) in ((5_000_000, 1_000_000, 150, 0.01, 31),
                                                                                 (100_000_000, 10_000_000, 150, 0.01, 31))
        label = (args.config if std else "custom") + ": " + \
            ("kbo find (max_gap_len=0; run lengths on the device)" if args.find else
             "kbo map (fill_gaps=false, call_variants=false, format=true)")
        a1_kernels = ("plan_kernel + plan_count/scan/emit + ms_walk_guided_kernel (ms_walk_recovery_kernel from 24 Mi rows on) + redo_collect + ms_walk_kernel (flagged reads)"
                      if planned else "ms_walk_kernel")
        result = {
            "metric": f"query Mbp/sec for kbo {'find' if args.find else 'map'}, k={args.k}, {args.genome / 1e6:g} Mbp SBWT; bit-exact MS vs CPU",
            "value": round(world * bases * args.steps / elapsed / 1e6, 1),
            "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{label}, {args.genome / 1e6:g} Mbp iid genome SBWT k={args.k}, "
                                   f"{args.reads} x {args.read_len} bp reads per GPU, "
                                   f"{args.sub_rate * 100:g}% substitutions",
                       "index_n_sets": sbwt.n_sets(), "threshold": dev.threshold,
                       "walk": "plan-guided (path cover + guided walk)" if planned else "plain",
                       "index_device_bytes": {"rank_blocks": rank_b, "lcs": lcs_b, "two_base_blocks": pair_b,
                                              "path_cover": plan_b},
                       "parallelism": f"index replicated x{world}, reads sharded, no collective"},
            "roofline": {
                # SURVEY.md 8(d)'s contract figure: algorithmic bytes of the REFERENCE algorithm (64 B per 512-bit rank
                # block it would touch + 1 B per LCS element + 2) over the time of the A1 stage.  It is not a
                # bandwidth utilisation: this stage loads 16-byte rank blocks, and skips the stretches of a read that
                # match the index's path cover, so frac can exceed 1.  What binds the stage is in `bound`;
                # traffic_frac is the measured fabric traffic over the same time against the same peak.
                "bound": ("l2-miss line fills, by their latency: the guided walk keeps 8-12 waves per CU so that the lines of the "
                          "lanes in flight stay in L2 (DESIGN.md sections 4.2, 6)") if planned else
                         (("l2-miss line fills (index is L2/Infinity-Cache resident: about 56 G fills/s on this part, "
                           "DESIGN.md section 6)") if resident else "l2-miss line fills from HBM (about 56 G fills/s, DESIGN.md section 6)"),
                "achieved": round(achieved, 1) if achieved is not None else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4) if achieved is not None else None,
                "frac_meaning": "reference-algorithm bytes / A1 stage time / 8 TB/s (contract figure, may exceed 1)",
                "traffic": traffic, "traffic_source": tsrc,
                "traffic_frac": round(traffic / (walk_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
                "l2_miss_per_base": round(misses / bases, 4) if misses else None,
                "kernel": "A1 stage = " + a1_kernels, "kernel_ms": round(walk_ms, 4),
                "kernel_ms_per_rank": {"min": round(min(walk_all), 4), "max": round(max(walk_all), 4)},
                "algorithmic_bytes_per_base": round(b_alg, 2) if b_alg is not None else None},
            "kernels_ms": {"a1_stage": round(walk_ms, 4), "derand_translate": round(dt_ms, 4),
                           **({"run_lengths": round(rle_ms, 4)} if args.find else {})},
            "cpu_baseline": cpu,
            "bit_exact_vs_oracle": exact,
            "reference_ops_per_base": ops,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
