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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime spreads a process's streams over 4 hardware queues by default.  This process has torch's current stream, the
# second-pass stream and the three stage streams of the host batch pipeline: with 4 queues the pipeline's upload, kernel and
# download streams share queues with each other and stop overlapping (kbo_map_batch 21 instead of 40 Gbp/s, packed 72 instead of
# 123: tools/dbg_h2h3.py).  Read by the runtime when it starts, so set before anything touches the GPU; a value given from outside wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
FILL_CEILING_PER_S = 56e9   # L2-miss line fills/s this part delivers to dependent 16-byte gathers from tables beyond L2:
FILL_CEILING_SOURCE = "profiles/r01_ubench_gather4.txt (tools/ubench/gather4.hip: 55-59 G loads/s for 67 MB .. 4.3 GB tables)"
PRESETS = {"C2": (5_000_000, 1_000_000, False, "weak"), "C3": (100_000_000, 10_000_000, True, "weak"),
           "C4": (250_000_000, 100_000_000, False, "strong")}
SLAB_READS = 8_000_000      # reads per device-resident slab (one launch covers < 4 GiB of query)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # (0.26 ms each: a timed region of 50 ms - one stall of the host does not decide the line)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(PRESETS) + ["C5"], default="C2",
                    help="BASELINE.json workload: C2 = kbo map, 5 Mbp index, 1 M x 150 bp reads per GPU (the metric config); "
                         "C3 = kbo find, 100 Mbp index, 10 M reads per GPU (SURVEY.md 8(d)'s designated roofline run); "
                         "C4 = kbo map, 250 Mbp index, 100 M reads sharded over the GPUs")
    ap.add_argument("--genome", type=int, default=None)
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (C4: reads in all)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub-rate", type=float, default=0.01)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall-clock budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip everything that runs the oracle (parity gate, CPU "
                    "baseline, stage model) and the sensitivity / host-to-host legs: profiling runs")
    ap.add_argument("--no-extras", action="store_true", help="skip the sensitivity and host-to-host legs only")
    ap.add_argument("--extras", action="store_true", help="run the sensitivity and host-to-host legs on a custom workload too "
                    "(they run by default on the preset configurations, on rank 0 at N = 1)")
    ap.add_argument("--find", action="store_true",
                    help="time kbo::find instead of kbo::map: the step ends with format::run_lengths on the device")
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--call", action="store_true",
                    help="time the first pass of kbo call (C5 shape, scaled): MS walk with intervals + the breakpoint scan "
                         "on the device over 10 kbp reads (defaults: --genome 100000000 --reads 10000 --read-len 10000)")
    ap.add_argument("--no-plan", action="store_true", help="plain walk kernel only (no path cover, no plan-guided walk)")
    ap.add_argument("--two-kernels", action="store_true", help="kbo_ms_batch_dev + kbo_derand_translate_dev instead of kbo_map_batch_dev "
                    "(the MS values of every base go through HBM)")
    ap.add_argument("--one-at-a-time", action="store_true", help="one stream: a batch's second pass before the next batch's kernel")
    ap.add_argument("--pipelines", type=int, default=2, help="pipelines of two streams (kernel; second pass) the steps go to in turn, "
                    "two resident batches each (1: round 4's first form, two batches in flight; 3 is slower than 2)")
    ap.add_argument("--depth-table", type=int, default=0,
                    help="order of the depth table (kbo_set_depth_table): 0 = by index size, -1 = none (units + guided walk)")
    ap.add_argument("--index-cache", default=None, help="index file (.kbohip, with its path cover) to load instead of building; "
                    "written first if it does not exist")
    args = ap.parse_args(argv)
    if args.config == "C5":  # kbo call, 3 Gbp index, k = 63, 1 M x 10 kbp reads over 8 GPUs: one GPU's share (125 k reads)
        args.call = True
        args.c5 = args.genome is None and args.reads is None
        args.genome = args.genome if args.genome is not None else 3_000_000_000
        args.reads = args.reads if args.reads is not None else 125_000
        args.k = args.k if args.k != 31 else 63
        if args.steps == 200:
            args.steps = 3
        args.warmup = min(args.warmup, 1)
    else:
        args.c5 = False
    if args.call:
        args.genome = args.genome if args.genome is not None else 100_000_000
        args.reads = args.reads if args.reads is not None else 10_000
        args.read_len = args.read_len if args.read_len != 150 else 10_000
        args.custom = True
        args.scaling = "weak"
        return args
    preset = PRESETS[args.config]
    args.custom = args.genome is not None or args.reads is not None
    args.genome = args.genome if args.genome is not None else preset[0]
    args.reads = args.reads if args.reads is not None else preset[1]
    args.find = args.find or (preset[2] and not args.custom)
    args.scaling = preset[3]
    return args


def build_sha16():
    """A fingerprint of what a profile was taken of: bench.py and every source of the HIP extension.  tools/profile_bench.sh stores
    it next to the counters it collects; a line printed by a different build quotes no traffic figure (a committed
    profile must not decorate the line of a later build)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(ROOT, "bench.py")] + sorted(glob.glob(os.path.join(ROOT, "kbo_amd", "csrc", "*.h*")) +
                                                       glob.glob(os.path.join(ROOT, "kbo_amd", "csrc", "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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


# ---------------------------------------------------------------------------------------------- index: build once, cache

def cache_path(args):
    if args.index_cache:
        return args.index_cache
    d = os.environ.get("KBO_BENCH_CACHE_DIR", "/tmp")
    return os.path.join(d, f"kbo_bench_iid_{args.genome}_k{args.k}{'_noplan' if args.no_plan else ''}.kbohip")


def build_or_load_index(args, threads, may_build=True):
    """-> (genome, sbwt).  The index file carries the path cover (kbo_index_save), so a rank that loads it uploads after a
    few streaming passes instead of repeating the build and the cover's pointer chase."""
    import kbo_amd
    from kbo_amd import index as kindex, synth
    genome = synth.genome(args.genome)
    path = cache_path(args)
    if os.path.exists(path):
        try:
            sbwt, _ = kindex.load_flat(path)
            if sbwt.k() == args.k and sbwt.n_kmers() > 0:
                return genome, sbwt
        except Exception as e:  # (a stale or torn file: build again)
            print(f"[bench] index cache {path} unusable ({e}); rebuilding", file=sys.stderr)
    if not may_build:
        raise SystemExit(f"bench.py: index cache {path} missing")
    sbwt, _ = kbo_amd.build([genome], kbo_amd.BuildOpts(k=args.k, num_threads=min(16, max(1, threads))))
    try:
        tmp = f"{path}.{os.getpid()}.tmp"
        kindex.save_flat(tmp, sbwt)  # (computes the cover while the plan is enabled)
        os.replace(tmp, path)
    except Exception as e:
        print(f"[bench] could not write the index cache {path}: {e}", file=sys.stderr)
    return genome, sbwt


def shard(args, rank, world):
    """-> (reads of this rank, index of its first read).  weak scaling (C2, C3): args.reads per rank; strong (C4):
    args.reads in all, contiguous ranges; the shards tile the read set exactly (tests/test_dist_gloo.py)."""
    if args.scaling == "strong":
        per_rank = (args.reads + world - 1) // world
        first = rank * per_rank
        return max(0, min(per_rank, args.reads - first)), first
    return args.reads, rank * args.reads


def spawn_command(args, argv, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if not args.index_cache:
        cmd += ["--index-cache", cache_path(args)]
    return cmd


def spawn_ranks(args, argv):
    """--gpus N > 1 from a plain `python bench.py`: build the cache here (host only), start N ranks, relay their output."""
    import kbo_amd
    if args.no_plan:
        kbo_amd.lib().kbo_set_plan(0, 0, 0)
    cores, _ = usable_cores()
    build_or_load_index(args, cores)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = spawn_command(args, argv, port)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------- oracle-side legs (rank 0)

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


def run_piped(devs, mstream, steps):
    """`steps` batches through the library's pipelines (kbo_hip.h kbo_map_stream_*: pairs of kernel stream and second-pass stream that
    take the batches in turn, two slots each), the resident batches in `devs` in turn -> their tickets"""
    return [mstream.submit(devs[i % len(devs)]) for i in range(steps)]


def run_batch(devs, stream, find, steps, warmup, torch, device, two_kernels=False, pipes=None):
    """warm-up + timed steps over the resident batches `devs` in turn (one, or two per pipeline of the same shape with `pipes`, see
    run_piped) -> (elapsed s, a1 ms, a5/a6 ms, rle ms | None); with kbo_map_batch_dev (not two_kernels) a1 = the whole step and a5/a6 = 0"""
    dev = devs[0]
    if not two_kernels:
        from kbo_amd import batch
        mstream = batch.MapStream(dev.sbwt, max(d.n_seqs for d in devs), max(d.total for d in devs), max(d.max_len for d in devs), pipelines=pipes) if pipes else None

        def go(n):
            if mstream is not None:
                return run_piped(devs, mstream, n)
            for i in range(n):
                devs[i % len(devs)].run(stream)
            return []
        go(warmup)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        tickets = go(steps)
        for t in tickets[-2 * (pipes or 1):]:  # (the event behind the timed batches: the slots that may still be busy)
            mstream.wait_on(t, stream)
        e1.record(stream)
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        if mstream is not None:
            mstream.close()
        return elapsed, e0.elapsed_time(e1) / steps, 0.0, None
    for _ in range(warmup):
        dev.run(stream)
        if find:
            dev.run_lengths(0, stream)
    torch.cuda.synchronize(device)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    t0 = time.perf_counter()
    for s in range(steps):
        ev[s][0].record(stream)
        dev.walk(stream)
        ev[s][1].record(stream)
        dev.derand_translate(stream)
        ev[s][2].record(stream)
        if find:
            dev.run_lengths(0, stream)
        ev[s][3].record(stream)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    return (elapsed, float(np.mean([e[0].elapsed_time(e[1]) for e in ev])), float(np.mean([e[1].elapsed_time(e[2]) for e in ev])),
            float(np.mean([e[2].elapsed_time(e[3]) for e in ev])) if find else None)


def indel_reads(genome, n_reads, L, sub_rate, indel_rate, seed, many=False):
    """reads of L bases with substitutions and, with probability 1 - (1 - indel_rate)^L per read, one insertion or deletion of
    1 - 3 bases at a random place (numpy; synth.reads makes substitutions only) -> (concat uint8, offsets uint64).
    many: long reads put together from pieces of 1 / indel_rate bases on average, a base dropped or a random one added between
    two pieces (one event per piece boundary)"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    if many:
        out = np.empty(n_reads * L, dtype=np.uint8)
        for r in range(n_reads):
            src = int(rng.integers(0, len(genome) - 2 * L))
            parts, have = [], 0
            while have < L:
                n = int(rng.geometric(indel_rate))
                parts.append(genome[src:src + n])
                src += n
                have += n
                if rng.random() < 0.5:
                    src += 1                                          # a deletion
                else:
                    parts.append(acgt[rng.integers(0, 4, 1)])           # an insertion
                    have += 1
            rd = np.concatenate(parts)[:L].copy()
            hit = rng.random(L) < sub_rate
            rd[hit] = acgt[(np.searchsorted(acgt, rd[hit]) + rng.integers(1, 4, int(hit.sum()))) % 4]
            out[r * L:(r + 1) * L] = rd
        return out, np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
    out = np.empty(n_reads * L, dtype=np.uint8)
    p_read = 1.0 - (1.0 - indel_rate) ** L
    for a in range(0, n_reads, 100_000):
        n = min(100_000, n_reads - a)
        start = rng.integers(0, len(genome) - L - 8, n)
        has = rng.random(n) < p_read
        pos = rng.integers(10, L - 10, n)
        size = rng.integers(1, 4, n)
        ins = rng.random(n) < 0.5
        i = np.arange(L)[None, :]
        # deletion of `size` bases at pos: bases from pos on come from further right; insertion: from further left behind it
        shift = np.where(has[:, None] & (i >= pos[:, None]), np.where(ins[:, None], -np.minimum(size[:, None], i - pos[:, None] + 0), size[:, None]), 0)
        reads = genome[start[:, None] + i + shift]
        new = has[:, None] & ins[:, None] & (i >= pos[:, None]) & (i < (pos + size)[:, None])  # the inserted bases
        reads = np.where(new, acgt[rng.integers(0, 4, (n, L))], reads)
        hit = rng.random((n, L)) < sub_rate
        reads = np.where(hit, acgt[(np.searchsorted(acgt, reads) + rng.integers(1, 4, (n, L))) % 4], reads)
        out[a * L:(a + n) * L] = reads.reshape(-1)
    return out, np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)


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
    # kbo_ms_batch_dev: the walk alone (plan + depth table + second pass inside it), MS bytes out
    dev = batch.DeviceBatch(sbwt, concat, offsets, device=device, format=True, want_ms=True)
    for _ in range(4):
        dev.walk(stream)
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
    del dev
    # kbo_map_batch_dev(want_ms): the one kernel in its MS-emitting form, MS bytes + formatted characters out
    devs = [batch.DeviceBatch(sbwt, concat, offsets, device=device, format=True, want_ms=True) for _ in range(2 * pipes if pipes else 1)]
    elapsed, _, _, _ = run_batch(devs, stream, False, 40, 8, torch, device, False, pipes)
    ok = all(bool(np.array_equal(d.ms[:total].cpu().numpy(), exp_d) and np.array_equal(d.chars[:total].cpu().numpy(), exp_map)) for d in devs)
    out["kbo_map_batch_dev_want_ms"] = {"value": round(total * 40 / elapsed / 1e6, 1), "unit": "Mbp/s", "steps": 40, "step_ms": round(elapsed / 40 * 1e3, 4),
                                        "bytes_out_per_base": 2, "one_kernel": bool(devs[0].fused), "batches_in_flight": len(devs),
                                        "bit_exact_vs_oracle": ok}
    return out


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
                            "the one kernel for sequences of any length (long_kernels.hip) needs a table"}
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


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
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
    pipes = [(stream if p == 0 else torch.cuda.Stream(device), torch.cuda.Stream(device)) for p in range(n_pipes)] if piped and args.find else None
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
    for w in range(args.warmup):
        one_step(w)
    sync_all()
    ev = [[[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in slabs] for _ in range(args.steps)]
    import ctypes as _C
    L.kbo_set_stage_timing(1 if one_kernel else 0)  # (event records per call: the dominant kernel's own duration, live)
    import gc
    gc.collect()
    gc.disable()  # (the timed region is tens of milliseconds of enqueueing: no collector pause inside it)
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(args.warmup + s, ev[s])
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
    serial = None
    if piped and rank == 0 and world == 1:  # the same steps on one stream, for the record
        n_ser = max(1, min(args.steps, 10))
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
        sens = h2h = ms_var = None
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
                "redo_pass": "redo_collect + ms_walk_kernel + derand_flagged over the %.2f %% of the reads the kernel leaves (a chain of dependent "
                             "look-ups: its time is the chain's, not the reads')" % (100.0 * c["tab_unresolved"] / dev.n_seqs),
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
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{label}, {args.genome / 1e6:g} Mbp iid genome SBWT k={args.k}, "
                                   + (f"{args.reads} x {args.read_len} bp reads in all, {n_mine} per GPU, " if args.scaling == "strong"
                                      else f"{args.reads} x {args.read_len} bp reads per GPU, ")
                                   + f"{args.sub_rate * 100:g}% substitutions",
                       "index_n_sets": sbwt.n_sets(), "threshold": dev.threshold,
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
            "host_to_host": h2h,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
