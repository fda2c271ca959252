#!/usr/bin/env python3
"""Sequences of more than 160 bases through kbo_map_batch_dev (long_kernels.hip) on the C2 index: throughput, what the pieces did,
every base of the first 3 Mbp against the oracle.  python tools/exp_long.py [--genome 5000000] [--len 10000] [--mbases 150]
[--steps 20] [--variants 1pct,ont,clean,5pct,big] [--no-check] [--tail]"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (the application asks for the hardware queues its streams need: INTEGRATION.md)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--len", type=int, default=10_000)
    ap.add_argument("--mbases", type=int, default=150)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--variants", default="1pct,ont")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--count", action="store_true", help="work counters of the kernel (kbo_set_plan_stats: slower)")
    ap.add_argument("--tail", action="store_true", help="the flagged pieces' pass on a second stream, two batches in flight")
    ap.add_argument("--pipes", type=int, default=0, help="the library's own pipelines (kbo_map_stream_*), two batches in flight each")
    args = ap.parse_args()
    import torch

    import bench
    import kbo_amd
    from kbo_amd import batch, synth
    from oracle import binding as ora
    dev0 = torch.device("cuda:0")
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    g = synth.genome(args.genome)
    t0 = time.perf_counter()
    sbwt, _ = kbo_amd.build([g], kbo_amd.BuildOpts(k=args.k, num_threads=cores))
    sbwt.to_device(-1)
    kbo_amd.lib().kbo_set_plan_stats(1 if args.count else 0)
    print("index built + on the device in %.1f s; depth table of %d bases" % (time.perf_counter() - t0, sbwt.depth_table_order()), flush=True)
    oi = None
    if not args.no_check:
        rows, Carr, lcs = sbwt.export_parts()
        oi = ora.Index.from_parts(args.k, sbwt.n_sets(), sbwt.n_kmers(), rows, Carr, lcs)
    n = max(1, args.mbases * 1_000_000 // args.len)
    stream = torch.cuda.Stream(dev0)
    tail = torch.cuda.Stream(dev0) if args.tail else None
    for name in args.variants.split(","):
        if name == "1pct":
            concat, offsets = synth.reads(g, n, args.len, 0.01, seed=0x5E11E)
        elif name == "clean":
            concat, offsets = synth.reads(g, n, args.len, 0.0, seed=0x5E11F)
        elif name == "5pct":
            concat, offsets = synth.reads(g, n, args.len, 0.05, seed=0x5E120)
        elif name == "ont":
            concat, offsets = bench.indel_reads(g, max(1, n // 2), args.len, 0.025, 0.025 / 2, seed=0x5E11D, many=True)
        elif name == "big":
            concat, offsets = bench.indel_reads(g, max(1, n // 2), args.len, 0.01, 0.002, seed=0x5E121, many=True)
        else:
            raise SystemExit("unknown variant " + name)
        devs = [batch.DeviceBatch(sbwt, concat, offsets, device=dev0, format=True, want_ms=False) for _ in range(2 * args.pipes if args.pipes else 2 if args.tail else 1)]
        if args.pipes:
            mstr = batch.MapStream(sbwt, devs[0].n_seqs, devs[0].total, 0, pipelines=args.pipes)
            for i in range(4):
                mstr.submit(devs[i % len(devs)])
            mstr.sync()
            t0 = time.perf_counter()
            for i in range(args.steps):
                mstr.submit(devs[i % len(devs)])
            mstr.sync()
            ms = (time.perf_counter() - t0) * 1e3 / args.steps
            total = int(offsets[-1])
            line = "%-6s %7.1f Gbp/s  %.3f ms per batch of %d Mbases (%d pipelines, host clock)" % (name, total / ms / 1e6, ms, total // 1_000_000, args.pipes)
            if oi is not None:
                n_chk = max(1, int(np.searchsorted(offsets, 3_000_000)))
                n_b = int(offsets[n_chk])
                exp = oi.matches_batch(concat[:n_b], offsets[:n_chk + 1], 1e-7, n_threads=cores)
                exp = np.frombuffer(ora.relative_to_ref(concat[:n_b], exp), dtype=np.uint8)
                line += "  bit-exact %s" % all(bool(np.array_equal(d.chars[:n_b].cpu().numpy(), exp)) for d in devs)
            print(line, flush=True)
            mstr.close()
            del devs
            continue
        with torch.cuda.stream(stream):
            for i in range(3):
                devs[i % len(devs)].run(stream, tail)
            torch.cuda.synchronize(dev0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(args.steps):
                devs[i % len(devs)].run(stream, tail)
            if tail is not None:
                stream.wait_stream(tail)
            e1.record(stream)
            torch.cuda.synchronize(dev0)
        ms = e0.elapsed_time(e1) / args.steps
        st = devs[0].long_stats(stream)
        total = int(offsets[-1])
        line = "%-6s %7.1f Gbp/s  %.3f ms per batch of %d Mbases; pieces %d flagged %d (%.2f %%) sub-items %d; per kb: seeds %.2f filter %.2f table %.2f second %.3f" % (
            name, total / ms / 1e6, ms, total // 1_000_000, st["pieces"], st["flagged"], 100.0 * st["flagged"] / max(1, st["pieces"]), st["sub_items"],
            1e3 * st["seed_lookups"] / total, 1e3 * st["filter_lookups"] / total, 1e3 * st["table_lookups"] / total, 1e3 * st["second_lookups"] / total)
        if st.get("band_tried"):
            line += "  band tried %d taken %d" % (st["band_tried"], st["band_taken"])
        if args.count:
            line += "  why: " + " ".join("%s %d" % (k_[4:], st[k_]) for k_ in st if k_.startswith("why_"))
        if st.get("cyc_staging"):
            line += "  cycles per piece: " + " ".join("%s %d" % (k_[4:], st[k_] // max(1, st["pieces"])) for k_ in st if k_.startswith("cyc_"))
        if oi is not None:
            n_chk = max(1, int(np.searchsorted(offsets, 3_000_000)))
            n_b = int(offsets[n_chk])
            exp = oi.matches_batch(concat[:n_b], offsets[:n_chk + 1], 1e-7, n_threads=cores)
            exp = np.frombuffer(ora.relative_to_ref(concat[:n_b], exp), dtype=np.uint8)
            line += "  bit-exact %s" % bool(np.array_equal(devs[0].chars[:n_b].cpu().numpy(), exp))
        print(line, flush=True)
        del devs


if __name__ == "__main__":
    main()
