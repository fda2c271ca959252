"""bench.py: arguments, presets, the index (built or loaded from the cache file), ranks, the loops that time resident batches,
synthetic reads with insertions and deletions."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # the repository: bench.py lives there
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
    ap.add_argument("--condition-ms", type=float, default=30.0,
                    help="untimed steps of the same workload in front of the warm-up steps, for so many milliseconds: the device comes out of the idle "
                         "set-up phase with low clocks and needs about 15 ms of load to be back at the rate a serving process sees (0: none)")
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
    ap.add_argument("--no-whole-call", action="store_true", help="--call / --config C5: the first pass only (profiler passes)")
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
    """A fingerprint of what a profile was taken of: bench.py with its parts (benchlib/) and every source of the HIP extension.  tools/profile_bench.sh stores
    it next to the counters it collects; a line printed by a different build quotes no traffic figure (a committed
    profile must not decorate the line of a later build)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(ROOT, "bench.py")] + sorted(glob.glob(os.path.join(ROOT, "benchlib", "*.py"))) + \
            sorted(glob.glob(os.path.join(ROOT, "kbo_amd", "csrc", "*.h*")) +
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
    if os.environ.get("KBO_BENCH_NO_CACHE") == "1":  # (a run that builds the index once and uses it once: tests)
        return genome, sbwt
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
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)
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



def run_piped(devs, mstream, steps):
    """`steps` batches through the library's pipelines (kbo_hip.h kbo_map_stream_*: pairs of kernel stream and second-pass stream that
    take the batches in turn, two slots each), the resident batches in `devs` in turn -> their tickets"""
    return [mstream.submit(devs[i % len(devs)]) for i in range(steps)]


CONDITION_MS = 30.0  # (bench.py sets it from --condition-ms)


def condition(go, warmup, torch, device):
    """The untimed steps in front of a timed region: go(n) enqueues n steps.  Four of them sized by the host's clock, then CONDITION_MS of
    them and the warm-up steps back to back - a device that has idled through host work (set-up, an oracle comparison) runs its first
    ~15 ms of load 5 - 10 % slower than the same work later (tools/ramp_timeline.py), and 8 warm-up steps are 1 - 2 ms"""
    if CONDITION_MS <= 0:
        go(warmup)
        return
    t0 = time.perf_counter()
    go(4)
    torch.cuda.synchronize(device)
    per_step = max(1e-6, (time.perf_counter() - t0) / 4)
    go(min(4000, int(CONDITION_MS * 1e-3 / per_step) + 1) + warmup)


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
        condition(go, warmup, torch, device)
        torch.cuda.synchronize(device)
        if mstream is not None:
            # (the host's clock between two synchronisations, as the headline's: events recorded on a torch stream beside the library's
            # own four streams put that stream on a hardware queue one of them uses - the same 40 batches of 10 kbp reads then took
            # 8 % longer with the events on the default stream, 60 % longer on a fresh one)
            t0 = time.perf_counter()
            go(steps)
            mstream.sync()
            elapsed = time.perf_counter() - t0
            mstream.close()
            return elapsed, elapsed / steps * 1e3, 0.0, None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        go(steps)
        e1.record(stream)
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
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


